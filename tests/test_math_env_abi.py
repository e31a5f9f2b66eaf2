"""CPU: deterministic math header, env restatements, Philox known answers, and the C-ABI surface of the HIP library."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from alphazero_gym_amd.envs import CartPoleEnv, MountainCarEnv, PendulumEnv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    lib = O.lib()
    lib.azo_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
    lib.azo_philox.restype = None
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        out = (C.c_uint32 * 4)()
        lib.azo_philox(*ctr, *key, out)
        assert tuple(out) == want


def test_math_accuracy_against_libm():
    x = np.linspace(-20, 5, 100001); xf = x.astype(np.float32).astype(np.float64)
    assert np.max(np.abs(O.math_eval(0, x) / np.exp(xf) - 1)) < 2e-7
    x = np.linspace(-20, 20, 100001); xf = x.astype(np.float32).astype(np.float64)
    ref = np.expm1(xf)
    assert np.max(np.abs(O.math_eval(1, x) - ref) / np.maximum(np.abs(ref), 1e-30)) < 3e-7
    assert O.math_eval(1, np.array([0.0]))[0] == 0.0 and not np.signbit(O.math_eval(1, np.array([-0.0]))[0])
    x = np.linspace(-12, 12, 100001); xf = x.astype(np.float32).astype(np.float64)
    assert np.max(np.abs(O.math_eval(2, x) - np.tanh(xf))) < 2e-7
    x = np.linspace(1e-8, 1, 100001)[:-1]; xf = x.astype(np.float32).astype(np.float64)
    assert np.max(np.abs(O.math_eval(3, x) - np.log(xf)) / np.maximum(np.abs(np.log(xf)), 1e-3)) < 1e-6
    assert np.max(np.abs(O.math_eval(4, x) - np.cos(2 * np.pi * xf))) < 2e-7
    x = np.linspace(-100, 100, 200001)
    assert np.max(np.abs(O.math_eval(5, x) - np.sin(x))) < 3e-16 and np.max(np.abs(O.math_eval(6, x) - np.cos(x))) < 3e-16
    np.testing.assert_array_equal(O.math_eval(7, x), x % (2 * np.pi))   # Python float % is exact; so is azg_pymod
    n = O.math_eval(8, np.arange(200000))
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01


def test_pendulum_c_env_matches_numpy_env():
    rng = np.random.Generator(np.random.PCG64(3))
    for version, env_id in ((0, 1), (1, 2)):
        for _ in range(300):
            th, thd = rng.uniform(-40, 40), rng.uniform(-8, 8)
            u = np.float32(rng.uniform(-3, 3))
            e = PendulumEnv(state=[th, thd], version=version)
            obs, r, done, _ = e.step(np.array([[u]], dtype=np.float32))
            nxt, rc, dc, obsc = O.env_step(env_id, [th, thd], u)
            np.testing.assert_allclose(nxt, e.state, rtol=0, atol=1e-13)
            assert abs(rc - float(r[0])) < 1e-13 and dc is False and done is False
            np.testing.assert_allclose(obsc, obs.astype(np.float32), atol=1e-7)


def test_cartpole_c_env_matches_numpy_env():
    rng = np.random.Generator(np.random.PCG64(4))
    n_done = 0
    for _ in range(400):
        s = rng.uniform(-1, 1, 4) * np.array([2.6, 3.0, 0.23, 3.0])
        a = int(rng.integers(0, 2))
        e = CartPoleEnv(state=s)
        obs, r, done, _ = e.step(a)
        nxt, rc, dc, obsc = O.env_step(0, s, a)
        np.testing.assert_allclose(nxt, e.state, rtol=0, atol=1e-13)
        assert rc == 1.0 == r and dc == done
        np.testing.assert_array_equal(obsc, obs)
        n_done += done
    assert 0 < n_done < 400


def test_mountaincar_c_env_matches_python_env():
    """gym MountainCar-v0 (three actions): the oracle's step against the Python restatement -- in the valley, at the inelastic left
    wall, at both speed limits and across the flag (done)."""
    rng = np.random.Generator(np.random.PCG64(5))
    n_done = n_wall = 0
    for i in range(600):
        s = np.array([rng.uniform(-1.2, 0.6), rng.uniform(-0.07, 0.07)])
        if i % 7 == 0:
            s = np.array([rng.uniform(-1.2, -1.19), rng.uniform(-0.07, 0.0)])    # into the wall
        if i % 11 == 0:
            s = np.array([rng.uniform(0.44, 0.5), rng.uniform(0.0, 0.07)])       # up to the flag
        a = int(rng.integers(0, 3))
        e = MountainCarEnv(state=s)
        obs, r, done, _ = e.step(a)
        nxt, rc, dc, obsc = O.env_step(3, s, a)
        np.testing.assert_allclose(nxt, e.state, rtol=0, atol=1e-15)
        assert rc == -1.0 == r and dc == done
        np.testing.assert_array_equal(obsc, obs)
        n_done += done
        n_wall += (nxt[0] == -1.2 and nxt[1] == 0.0)
    assert n_done > 10 and n_wall > 10
    for ep in range(5):
        s = O.reset_state(9, 3, ep, False, env_id=3)
        assert -0.6 <= s[0] <= -0.4 and s[1] == 0.0


def test_mountaincar_continuous_c_env_matches_python_env():
    """gym MountainCarContinuous-v0: the oracle's step against the Python restatement -- in the valley, at the inelastic left wall,
    at both speed limits, with actions beyond the [-1, 1] force clip (the reward uses the action as it came) and across the flag at
    0.45 (done, +100); the VecMountainCarContinuous batch form agrees with both."""
    from alphazero_gym_amd.envs import MountainCarContinuousEnv, VecMountainCarContinuous
    rng = np.random.Generator(np.random.PCG64(6))
    n_done = n_wall = 0
    S, A, NXT, R, D = [], [], [], [], []
    for i in range(600):
        s = np.array([rng.uniform(-1.2, 0.6), rng.uniform(-0.07, 0.07)])
        if i % 7 == 0:
            s = np.array([rng.uniform(-1.2, -1.19), rng.uniform(-0.07, 0.0)])    # into the wall
        if i % 5 == 0:
            s = np.array([rng.uniform(0.38, 0.45), rng.uniform(0.0, 0.07)])      # up to the flag
        a = np.float32(rng.uniform(-2.5, 2.5))
        e = MountainCarContinuousEnv(state=s)
        obs, r, done, _ = e.step(np.array([a], dtype=np.float32))
        nxt, rc, dc, obsc = O.env_step(4, s, a)
        np.testing.assert_allclose(nxt, e.state, rtol=0, atol=1e-15)
        assert r.shape == (1,) and r.dtype == np.float64 and rc == float(r[0]) and dc == done
        assert rc == (100.0 if done else 0.0) - float(a) * float(a) * 0.1
        np.testing.assert_array_equal(obsc, obs.astype(np.float32))
        n_done += done
        n_wall += (nxt[0] == -1.2 and nxt[1] == 0.0)
        S.append(s); A.append(a); NXT.append(nxt); R.append(rc); D.append(done)
    assert n_done > 30 and n_wall > 10
    v = VecMountainCarContinuous(len(S))
    v.state = np.array(S)
    rv, dv = v.step(np.array(A))
    np.testing.assert_allclose(v.state, np.array(NXT), rtol=0, atol=1e-15)   # (numpy's vector cos and libm's differ by an ulp now and then)
    np.testing.assert_array_equal(rv, np.array(R))
    np.testing.assert_array_equal(dv, np.array(D))
    for ep in range(5):
        s = O.reset_state(9, 3, ep, False, env_id=4)
        assert -0.6 <= s[0] <= -0.4 and s[1] == 0.0


def test_acrobot_c_env_matches_python_env():
    """gym Acrobot-v1: the oracle's step (include/azg_math.h: one classical Runge-Kutta step of the "book" dynamics, angle wrap, velocity
    bounds) against the Python restatement over the whole state space, all three torques, both sides of the terminal line; the
    observation is (cos, sin) of both angles + both velocities; reward -1, 0 on the step that ends the episode."""
    from alphazero_gym_amd.envs import AcrobotEnv
    rng = np.random.Generator(np.random.PCG64(7))
    n_done = n_wrap = 0
    for i in range(1500):
        s = np.array([rng.uniform(-np.pi, np.pi), rng.uniform(-np.pi, np.pi), rng.uniform(-4 * np.pi, 4 * np.pi), rng.uniform(-9 * np.pi, 9 * np.pi)])
        if i % 3 == 0:
            s[2:] *= 0.1                                              # (slow states: no wrap, no velocity bound)
        a = int(rng.integers(0, 3))
        e = AcrobotEnv(state=s)
        obs, r, done, _ = e.step(a)
        nxt, rc, dc, obsc = O.env_step(5, s, a)
        # (the Runge-Kutta step amplifies the <= 1 ulp differences between azg_sincos and libm at high angular velocities)
        np.testing.assert_allclose(nxt, e.state, rtol=0, atol=5e-13)
        assert rc == r == (0.0 if done else -1.0) and dc == done
        np.testing.assert_allclose(obsc, obs.astype(np.float32), rtol=0, atol=2e-7)
        assert obsc.shape == (6,)
        n_done += done
        n_wrap += abs(s[0] + 0.2 * s[2]) > np.pi
    assert n_done > 100 and n_wrap > 50
    for ep in range(5):
        s = O.reset_state(9, 3, ep, False, env_id=5)
        assert s.shape == (4,) and (np.abs(s) <= 0.1).all()


def test_hip_library_exports_every_symbol_of_the_header():
    """The drop-in boundary: every entry point declared in include/azgym.h must be exported by libazgym_hip.so
    (loading needs no GPU; no compute is called)."""
    hdr = open(os.path.join(ROOT, "include", "azgym.h")).read()
    names = sorted(set(re.findall(r"\b(azg_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 20
    path = os.path.join(ROOT, "alphazero_gym_amd", "csrc", "libazgym_hip.so")
    assert os.path.exists(path), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(path)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.azg_abi_version() == 1
    olib = O.lib()
    for n in names:
        if n in ("azg_math_selftest", "azg_search_info", "azg_debug_stamps"):   # (about the device: kernel forms, LDS residency, stamps)
            continue
        assert hasattr(olib, "azo_" + n[4:]), n


def test_product_package_never_touches_the_oracle():
    """Parity claims are void if the product path can route through the oracle: no module of the package may name it."""
    pkg = os.path.join(ROOT, "alphazero_gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in src and "libazg_oracle" not in src and "azo_" not in src.replace('prefix ``azo_``', ""), f


def _build_c_demo(tmp_path):
    """examples/c_abi_demo.c against include/azgym.h and libazgym_hip.so with plain gcc (C11): the header is C, the boundary needs
    neither Python nor torch."""
    import subprocess
    exe = os.path.join(str(tmp_path), "c_abi_demo")
    libdir = os.path.join(ROOT, "alphazero_gym_amd", "csrc")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"),
           "-o", exe, "-L" + libdir, "-lazgym_hip", "-Wl,-rpath," + libdir, "-lm"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


def test_c_program_builds_against_the_abi(tmp_path):
    _build_c_demo(tmp_path)


@pytest.mark.gpu
def test_c_program_runs_a_search_through_the_abi(tmp_path):
    import subprocess
    p = subprocess.run([_build_c_demo(tmp_path)], capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and "0 trees with a wrong visit total" in p.stdout, (p.stdout[-800:], p.stderr[-800:])


def test_loading_the_engine_before_torch_is_reported_not_left_to_hang():
    """a process that maps a HIP runtime before importing torch (an embedding application loading
    libazgym_hip.so first) would make `import torch` bring a second libamdhip64 -- torch.cuda then finds no GPU or hangs.
    _native.lib() raises HipRuntimeConflict instead; the C entry point reports two mapped runtimes through azg_last_error."""
    import subprocess
    import sys
    code = r'''
import ctypes, os, sys
sys.path.insert(0, %r)
from alphazero_gym_amd import _native, _capi
assert "torch" not in sys.modules
lib = ctypes.CDLL(_native.LIB_PATH)                     # what an embedding application would do: the engine first
assert len(_native.mapped_hip_runtimes()) == 1
try:
    _native.lib()
    print("NO ERROR")
except _native.HipRuntimeConflict as ex:
    print("CONFLICT", "Import torch before" in str(ex))
import torch                                            # now the second runtime is mapped as well
assert len(_native.mapped_hip_runtimes()) == 2
cfg = _capi.AzgConfig(); cfg.struct_size = ctypes.sizeof(_capi.AzgConfig); cfg.n_trees = 1; cfg.n_sims = 1
h = ctypes.c_void_p()
lib.azg_engine_create.restype = ctypes.c_int
lib.azg_last_error.restype = ctypes.c_char_p
rc = lib.azg_engine_create(ctypes.byref(cfg), ctypes.byref(h))
print("CREATE", rc, b"two HIP runtimes" in lib.azg_last_error(None))
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "CONFLICT True" in out.stdout, out.stdout + out.stderr
    assert "CREATE -3 True" in out.stdout, out.stdout + out.stderr


def test_only_the_checkers_touch_the_oracle():
    """The oracle is test infrastructure: besides tests/, only bench.py (its cpu_baseline leg) and __graft_entry__.py (smoke) may
    import or load it -- no tool, example or package module."""
    import glob
    offenders = []
    for pat in ("tools/**/*.py", "tools/**/*.sh", "examples/**/*.py", "alphazero_gym_amd/**/*.py"):
        for f in glob.glob(os.path.join(ROOT, pat), recursive=True):
            txt = open(f).read()
            if "oracle_lib" in txt or "libazg_oracle" in txt or "OracleEngine" in txt or "import test_hip_parity" in txt:
                offenders.append(os.path.relpath(f, ROOT))
    assert not offenders, offenders


def test_oracle_under_address_and_undefined_behaviour_sanitizers():
    """the C oracle built with -fsanitize=address,undefined (make -C oracle asan) runs oracle-backed tests clean.  Here a
    quick subset in a child interpreter with the sanitizer runtimes preloaded (the whole set: make -C oracle asan-test, ~30 s)."""
    import subprocess
    import sys
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    pre = ":".join(subprocess.check_output(["gcc", f"-print-file-name={n}"], text=True).strip() for n in ("libasan.so", "libubsan.so"))
    env = dict(os.environ, AZG_ORACLE_ASAN="1", LD_PRELOAD=pre, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_abi_errors.py"), os.path.join(ROOT, "tests", "test_properties.py")],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "passed" in out.stdout and "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
