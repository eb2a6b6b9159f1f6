// headless — Python-free driver of libhideseek.so through the C ABI.
// Counterpart of the reference's src/headless.cpp:24-103:  headless {CPU,CUDA} NUM_WORLDS NUM_STEPS [--rand-actions]
// prints FPS = worlds * steps / seconds.  ("CUDA" = GPU/HIP here; "CPU" is refused: there is no CPU path.)
// Build:  g++ -O2 -I include marl-hideandseek_amd/tools/headless.cpp -L marl-hideandseek_amd/lib -lhideseek \
//             -Wl,-rpath,'$ORIGIN/../lib' -o marl-hideandseek_amd/lib/headless
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include <hideseek.h>

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "%s TYPE(CPU|CUDA) NUM_WORLDS NUM_STEPS [--rand-actions]\n", argv[0]);
        return 1;
    }
    hs_config cfg{};
    cfg.exec_mode = std::strcmp(argv[1], "CUDA") == 0 ? HS_EXEC_GPU : HS_EXEC_CPU;
    cfg.gpu_id = 0;
    cfg.num_worlds = std::atoi(argv[2]);
    const int num_steps = std::atoi(argv[3]);
    const bool rand_actions = argc > 4 && std::strcmp(argv[4], "--rand-actions") == 0;
    cfg.sim_flags = HS_FLAG_DEFAULT;
    cfg.rand_seed = 5;
    cfg.min_hiders = cfg.max_hiders = 3;                 // src/headless.cpp:57-69
    cfg.min_seekers = cfg.max_seekers = 2;
    cfg.num_pbt_policies = 1;
    hs_sim *sim = nullptr;
    if (hs_create(&cfg, &sim) != HS_OK) { std::fprintf(stderr, "hs_create: %s\n", hs_last_error()); return 2; }
    if (hs_init(sim) != HS_OK) { std::fprintf(stderr, "hs_init: %s\n", hs_last_error()); return 2; }
    std::random_device rd;
    std::mt19937 rng(rd());
    std::uniform_int_distribution<int32_t> act(0, 4);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < num_steps; ++i) {
        if (rand_actions) {
            // src/headless.cpp:79-93: two Manager::setAction pokes per world with values 0-4 and g = l = 0; the agent
            // index is the reference's own `j * k` (agent 0 for k = 0, agent j for k = 1), kept as it is
            for (int j = 0; j < cfg.num_worlds; ++j)
                for (int k = 0; k < 2; ++k) {
                    const int32_t x = act(rng), y = act(rng), r = act(rng);
                    if (hs_set_action(sim, j * k, x, y, r, 0, 0) != HS_OK) { std::fprintf(stderr, "hs_set_action: %s\n", hs_last_error()); return 2; }
                }
        }
        if (hs_step(sim) != HS_OK) { std::fprintf(stderr, "hs_step: %s\n", hs_last_error()); return 2; }
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("FPS %f\n", (double)num_steps * cfg.num_worlds / sec);
    hs_destroy(sim);
    return 0;
}
