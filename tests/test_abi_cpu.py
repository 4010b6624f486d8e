"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/inversus_hip.h declares (no compute calls without a GPU), and the host-side
mirror of the reference interface behaves like the reference where no GPU is involved."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "inversus_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(inv_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("inv_create", "inv_destroy", "inv_seed", "inv_reset", "inv_step", "inv_render_obs",
                 "inv_get_state", "inv_set_state", "inv_gae", "inv_rollout", "inv_check", "inv_last_error"):
        assert must in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    from inversus_reinforcement_learning_amd import _build, _lib
    path = _build.build_library()
    assert os.path.exists(path)
    raw = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(raw, name), f"{name} declared in include/inversus_hip.h but not exported"
    lib = _lib.load()
    assert set(_lib.SIGNATURES) == set(declared_symbols())
    import re
    with open(os.path.join(ROOT, "include", "inversus_hip.h")) as fh:
        assert lib.inv_abi_version() == int(re.search(r"#define\s+INV_ABI_VERSION\s+(\d+)", fh.read()).group(1)) == 8


def test_code_object_is_gfx950():
    from inversus_reinforcement_learning_amd import _build
    blob = open(_build.build_library(), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_"):
        assert other not in blob


def test_discrete_to_action_table():
    # tests/test_rl_env_wrapper.py:25-45 of the reference
    from inversus_reinforcement_learning_amd.env_wrappers import discrete_to_action
    from inversus_reinforcement_learning_amd import ActionType, Direction
    assert discrete_to_action(0).type == ActionType.NONE and discrete_to_action(0).direction is None
    order = [Direction.UP, Direction.RIGHT, Direction.DOWN, Direction.LEFT]
    for base, typ in ((1, ActionType.MOVE), (5, ActionType.SHOOT), (9, ActionType.CHARGE_SHOOT)):
        for k, d in enumerate(order):
            a = discrete_to_action(base + k)
            assert a.type == typ and a.direction == d
    for bad in (-1, 13, 100):
        with pytest.raises(ValueError):
            discrete_to_action(bad)


def test_action_record_validates_like_the_reference():
    from inversus_reinforcement_learning_amd import Action, ActionType, Direction
    with pytest.raises(ValueError):
        Action(ActionType.MOVE, None)
    with pytest.raises(ValueError):
        Action(ActionType.NONE, Direction.UP)


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from inversus_reinforcement_learning_amd.env_wrappers import MultiEnvRunner
    with pytest.raises(RuntimeError):
        MultiEnvRunner(4, seed=0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "inversus_reinforcement_learning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "inv_oracle" not in src, f


def test_algorithmic_bytes_match_survey():
    import bench
    assert bench.algorithmic_bytes_per_env_step() == 7369            # SURVEY.md 8(d), 15x10 K=16 vs_dummy
    assert bench.algorithmic_bytes_per_env_step(selfplay=True) == 14585
    assert bench.algorithmic_bytes_per_env_step(32, 32, 16) == 49539


def test_graft_entry_build_runs():
    """The driver's "does it build" check: compiles (or finds up to date) the HIP library and the oracle, loads the
    library and checks the ABI version against the header."""
    import __graft_entry__ as g
    g.build()


def test_the_c_example_of_integration_md_compiles_against_the_header(tmp_path):
    """INTEGRATION.md section C (the C / C++ caller) is compiled as written -- as C99 by gcc and as C++ by g++ -- against
    include/inversus_hip.h: the documented call sequence cannot drift from the declared signatures."""
    import shutil
    import subprocess
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## C. From C / C++"):]
    block = re.findall(r"```c\n(.*?)```", sec, flags=re.S)
    assert block, "INTEGRATION.md section C holds a C block"
    lines = block[0].strip().splitlines()
    assert lines[0].startswith("#include")
    body = "\n    ".join(lines[1:])
    src = ("#include <stdio.h>\n#include <stdint.h>\n" + lines[0] + "\n"
           "int integration_example(const int64_t *seeds_host, float *d_obs, float *d_extra, float *d_reward, uint8_t *d_done,\n"
           "                        const uint8_t *d_a1, void *stream, int T)\n{\n    " + body + "\n    return inv_destroy(h);\n}\n")
    for cc, ext, std in (("gcc", "c", "-std=c99"), ("g++", "cpp", "-std=c++20")):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not found")
        f = tmp_path / f"example.{ext}"
        f.write_text(src)
        r = subprocess.run([cc, std, "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(f)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_the_build_line_of_integration_md_lists_every_source_and_flag_of_the_recipe():
    """INTEGRATION.md section C quotes the hipcc command; it must name exactly the sources and the flags `_build.py` compiles
    (round 4's verdict found the line one source short: a library built from it would not have loaded)."""
    from inversus_reinforcement_learning_amd import _build
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"Build: `(hipcc .*?)`", doc, flags=re.S)
    assert m, "INTEGRATION.md quotes the build command"
    words = m.group(1).split()
    srcs = sorted(w for w in words if w.endswith(".hip"))
    assert srcs == sorted(os.path.relpath(p, ROOT) for p in _build.SOURCES)
    for flag in _build.FLAGS:
        if not flag.startswith("-W"):
            assert flag in words, flag
