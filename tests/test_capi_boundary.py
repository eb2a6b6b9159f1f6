"""The C-ABI library loads without a GPU, exports every symbol include/hideseek.h declares, and
fails loudly (status code, no fallback) when it cannot run on a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "hideseek.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hs_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported(hideseek_lib):
    L = C.CDLL(hideseek_lib)
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/hideseek.h but not exported"
    L.hs_version.restype = C.c_char_p
    assert b"gfx950" in L.hs_version()


def test_cpu_exec_mode_is_refused_not_emulated(hideseek_lib):
    import gpu_hideseek
    with pytest.raises(NotImplementedError):
        gpu_hideseek.HideAndSeekSimulator(
            exec_mode=gpu_hideseek.madrona.ExecMode.CPU, gpu_id=0, num_worlds=4, sim_flags=0, rand_seed=0,
            min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)


def test_invalid_configs_raise(hideseek_lib):
    import gpu_hideseek
    kw = dict(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=4, sim_flags=0, rand_seed=0,
              min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
    for bad in (dict(num_worlds=0), dict(max_hiders=4), dict(min_seekers=3), dict(max_hiders=0, max_seekers=0,
                                                                               min_hiders=0, min_seekers=0)):
        with pytest.raises(ValueError):
            gpu_hideseek.HideAndSeekSimulator(**{**kw, **bad})


def test_no_gpu_means_error_not_fallback(hideseek_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import gpu_hideseek
    with pytest.raises(RuntimeError):
        gpu_hideseek.HideAndSeekSimulator(
            exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=4, sim_flags=0, rand_seed=0,
            min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)


def test_python_face_names_match_the_reference_binding():
    """src/bindings.cpp:20-121 — names the scripts touch."""
    import gpu_hideseek as g
    assert int(g.SimFlags.Default) == 0 and int(g.SimFlags.UseFixedWorld) == 1
    assert int(g.SimFlags.IgnoreEpisodeLength) == 2 and int(g.SimFlags.RandomFlipTeams) == 4
    assert int(g.SimFlags.ZeroAgentVelocity) == 8
    assert int(g.SimFlags.RandomFlipTeams | g.SimFlags.UseFixedWorld | g.SimFlags.ZeroAgentVelocity) == 13
    assert int(g.madrona.ExecMode.CPU) == 0 and int(g.madrona.ExecMode.CUDA) == 1
    getters = ["reset", "done", "prep_counter", "action", "reward", "self_data", "self_type", "self_mask",
               "agent_data", "box_data", "ramp_data", "visible_agents_mask", "visible_boxes_mask",
               "visible_ramps_mask", "global_positions", "depth", "rgb", "lidar", "seed", "ckpt_ctrl", "ckpt"]
    for n in getters + ["agent_mask"]:
        assert callable(getattr(g.HideAndSeekSimulator, n + "_tensor"))
    for n in ("init", "step", "jax"):
        assert callable(getattr(g.HideAndSeekSimulator, n))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "marl-hideandseek_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "hs_ref" not in txt and "oracle/" not in txt.replace("CPU oracle", ""), os.path.join(dp, f)
