"""Known answers for the game logic of src/sim.cpp, on the CPU oracle."""
import numpy as np


def test_episode_timeline_prep_reward_done(oracle):
    s = oracle.RefSim(4, rand_seed=0)
    s.init()
    A = s.A
    assert (s.tensor("prep_counter") == 96).all()                      # sim.cpp:461-464 at step 0
    rewards, dones, preps = [], [], []
    for i in range(245):
        s.step()
        rewards.append(s.tensor("reward").copy()); dones.append(s.tensor("done").copy())
        preps.append(s.tensor("prep_counter").copy())
    rewards, dones, preps = map(np.array, (rewards, dones, preps))
    # step() number i (0-based) runs rewards with curEpisodeStep == i, then observations with i+1
    assert (rewards[:95] == 0).all()                                   # reward 0 while cur_step < 95 (sim.cpp:822-825)
    assert (np.abs(rewards[95:240]) >= 1).all()
    assert (dones[:239] == 0).all() and (dones[239] == 1).all()        # done on the 240th step (sim.cpp:825-827)
    assert (dones[240] == 0).all()                                     # cleared by the next episode's first step
    assert (preps[:96, 0, 0] == np.arange(95, -1, -1)).all()           # 96 - step
    assert (preps[96:239] == 0).all() and (preps[239] == 96).all()     # new episode after the reset
    stype = s.tensor("self_type").reshape(-1)
    r = rewards[120].reshape(-1)
    inb = np.abs(r) == 1
    # seekers get the negated hider reward (sim.cpp:829-832)
    for w in range(4):
        rows = slice(w * A, (w + 1) * A)
        rr, tt, ok = r[rows], stype[rows], inb[rows]
        if ok.all():
            assert len(set((rr * np.where(tt == 1, 1, -1)).tolist())) == 1


def test_zero_force_actions_keep_agents_at_rest(oracle):
    """a=5 -> 0 N in the default mode (sim.cpp:221-223); a=2 -> 0 N with ZeroAgentVelocity (:248-250)."""
    for flags, neutral in ((0, 5), (8, 2)):
        s = oracle.RefSim(8, sim_flags=flags, rand_seed=1)
        s.init()
        b0, _ = s.bodies()
        for _ in range(30):
            s.tensor("action")[:] = [neutral, neutral, neutral, 0, 0]
            s.step()
        b1, m = s.bodies()
        agents = m[:, 11:, 0] >= 0
        d = np.linalg.norm(b1[:, 11:, :2] - b0[:, 11:, :2], axis=-1)[agents]
        # no drive: agents stay put, except those spawned overlapping something after 20 rejected
        # placements (level_gen.cpp:146), which are pushed out at <= 3 m/s
        assert (d < 1e-3).mean() > 0.7 and d.max() < 3.2


def test_seekers_frozen_during_prep_and_actions_consumed(oracle):
    s = oracle.RefSim(16, rand_seed=2)
    s.init()
    b0, _ = s.bodies()
    stype = s.tensor("self_type").reshape(16, s.A)
    for _ in range(60):
        s.tensor("action")[:] = [10, 10, 5, 0, 0]                       # +60 N in x and y
        s.step()
    b1, _ = s.bodies()
    act = s.tensor("action").reshape(16, s.A, 5)
    moved = np.linalg.norm(b1[:, 11:15, :2] - b0[:, 11:15, :2], axis=-1)
    # hiders act from step 0 and have their actions consumed to {2,2,2,0,0} (sim.cpp:365-369)
    assert (act[stype == 1] == [2, 2, 2, 0, 0]).all()
    # seekers return early while curEpisodeStep < 95 (sim.cpp:206-209, 276-279): not consumed, not driven
    assert (act[stype == 0] == [10, 10, 5, 0, 0]).all()
    assert np.median(moved[stype == 1]) > 5 * np.median(moved[stype == 0]) + 0.01


def test_out_of_bounds_penalty(oracle):
    """-10 when |x| or |y| >= 18 (sim.cpp:836): drive an agent of debug level 5 out of the arena."""
    s = oracle.RefSim(1, sim_flags=2 | 8, rand_seed=0, min_hiders=1, max_hiders=1, min_seekers=1, max_seekers=1)
    s.tensor("reset")[:] = 5                                           # one hider at the origin on a plane
    s.init()
    got_penalty = False
    for i in range(400):
        s.tensor("action")[0] = [4, 2, 2, 0, 0]                         # +800 N in x, ZeroAgentVelocity mode
        s.step()
        x = s.tensor("self_data")[0, 0]
        r = s.tensor("reward")[0, 0]
        if i >= 96 and abs(x) >= 18:
            assert r == 1.0 - 10.0
            got_penalty = True
            break
    assert got_penalty


def test_global_positions_tail_quirk_and_obs_zero_fill(oracle):
    s = oracle.RefSim(8, rand_seed=0, min_hiders=1, max_hiders=3, min_seekers=1, max_seekers=3)
    s.init()
    _, info = s.walls()
    box = s.tensor("box_data")
    agent = s.tensor("agent_data")
    mask = s.tensor("self_mask").reshape(8, s.A)
    for w in range(8):
        nb = info[w, 2]
        na = info[w, 4] + info[w, 5]
        for a in range(na):
            row = w * s.A + a
            assert (box[row, nb:] == 0).all() and (box[row, :nb, 12:15] != 0).all()     # sim.cpp:489-492
            assert (agent[row, na - 1:] == 0).all()                                      # sim.cpp:530-533
        assert mask[w, :na].all() and not mask[w, na:].any()


def test_lock_and_grab_happen_and_show_in_observations(oracle):
    """actionSystem (sim.cpp:270-370): with g/l pressed every step some agent eventually locks and grabs."""
    s = oracle.RefSim(64, sim_flags=8, rand_seed=5, min_hiders=3, max_hiders=3, min_seekers=3, max_seekers=3)
    s.init()
    rng = np.random.default_rng(0)
    saw_grab = saw_lock = False
    for i in range(200):
        a = np.stack([rng.integers(0, 5, 64 * 6), rng.integers(0, 5, 64 * 6), rng.integers(0, 5, 64 * 6),
                      np.ones(64 * 6, np.int64) * (i % 7 == 0), np.ones(64 * 6, np.int64) * (i % 11 == 0)], 1)
        s.tensor("action")[:] = a
        s.step()
        saw_grab |= bool(s.tensor("self_data")[:, 12].any())
        saw_lock |= bool(s.tensor("box_data")[:, :, 15:17].any() or s.tensor("ramp_data")[:, :, 12:14].any())
    assert saw_grab and saw_lock
    _, m = s.bodies()
    assert set(np.unique(m[:, :11, 1]).tolist()) <= {0, 2}
