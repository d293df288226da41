"""Property tests of the HOST plan compilers (mshgnn_plan.hpp, mshgnn_gen_plan.hpp) -- they emit the index tables every kernel trusts.
CPU only: libmshgnn_hostplan.so is a g++ build of the two headers (no HIP); the same driver also runs under AddressSanitizer + UBSan
(`make -C morphsym_hgnn_amd/csrc asan-host`).  Topologies come from hypothesis: 1-40 nodes per type, random relations, 1-8 layers,
hidden 128-2048.  Checked: every table index inside what it addresses (the library's own bounds walk, check=1), node liveness == the Python
mirror `spec.node_liveness` (what `ddp.flat_data_parallel(live_only=True)` relies on), descriptors the LDS-resident engine refuses are refused
with a message and taken by the generic engine, and the committed compile-time programs (mshgnn_spec_tables.inc) equal what the compiler emits now."""
import ctypes as C
import os
import subprocess
import sys

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from morphsym_hgnn_amd import engine as eng
from morphsym_hgnn_amd.spec import ModelSpec
from morphsym_hgnn_amd.topology import RobotTopology

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "morphsym_hgnn_amd", "csrc")
HPM = dict(OK=0, L=1, NN=2, NMLP=3, FUSED=4, SLAB=5, SL_HB=6, SL_BLK=7, FS_BLK=8, NTABLES=9, LIVE=80, NEED=112, COUNT=160)
PAIRS = (("base", "joint"), ("joint", "base"), ("joint", "joint"), ("foot", "joint"), ("joint", "foot"), ("base", "base"))


def _lib(asan=False):
    name = "libmshgnn_hostplan_asan.so" if asan else "libmshgnn_hostplan.so"
    path = os.path.join(ROOT, "morphsym_hgnn_amd", name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", CSRC, "asan-host" if asan else "../libmshgnn_hostplan.so"], check=True, capture_output=True)
    lib = C.CDLL(path)
    lib.mshgnn_hostplan_compile.argtypes = [C.POINTER(eng.MshgnnDesc), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.c_int]
    lib.mshgnn_hostplan_compile_gen.argtypes = [C.POINTER(eng.MshgnnDesc), C.POINTER(C.c_int32), C.c_int]
    lib.mshgnn_hostplan_last_error.restype = C.c_char_p
    return lib


@st.composite
def topologies(draw, max_nodes=40):
    n = {"base": draw(st.integers(1, min(4, max_nodes))), "joint": draw(st.integers(1, max_nodes)), "foot": draw(st.integers(1, max_nodes))}
    rels = []
    for s_, d_ in PAIRS:
        if s_ == d_ == "base" and not draw(st.booleans()):
            continue
        lo = 1 if d_ == "foot" else 0      # the decoder's type needs an in-edge for anything to be live
        k = draw(st.integers(lo, min(48, max(n[s_], n[d_]) * 2)))
        pairs = [[draw(st.integers(0, n[s_] - 1)), draw(st.integers(0, n[d_] - 1))] for _ in range(k)]
        rels.append(((s_, "connect", d_), pairs))
    layers = draw(st.integers(1, 8))
    hidden = 128 * draw(st.sampled_from([1, 1, 1, 2, 4, 8, 16]))
    return n, rels, layers, hidden


def make_spec(n, rels, layers, hidden):
    return ModelSpec(kind="mi", topology=RobotTopology(name="hyp", num_nodes=n, relations=rels), hidden=hidden, num_layers=layers,
                     widths={"base": 6, "joint": 5, "foot": 3}, regression=True, grf_dimension=1, group=None, num_timesteps=1)


def check_case(lib, n, rels, layers, hidden, dtype):
    """One descriptor through the host compilers; returns which engine took it."""
    spec = make_spec(n, rels, layers, hidden)
    holder = eng._DescHolder(spec, eng.DTYPE_CODES[dtype])
    meta = (C.c_int32 * HPM["COUNT"])()
    rc = lib.mshgnn_hostplan_compile(C.byref(holder.desc), None, 0, meta, 1)
    assert rc != -2, lib.mshgnn_hostplan_last_error().decode()      # a table entry out of range
    if rc >= 0:
        assert hidden == 128
        assert meta[HPM["L"]] == layers and meta[HPM["NN"]] == sum(n.values())
        live, need = spec.node_liveness()
        base = {"base": 0, "joint": n["base"], "foot": n["base"] + n["joint"]}
        for l in range(layers):
            for arr, off in ((live, HPM["LIVE"]), (need, HPM["NEED"])):
                want = sum(1 << (base[t] + i) for t in arr[l] for i in arr[l][t])
                got = (meta[off + 2 * l] & 0xFFFFFFFF) | ((meta[off + 2 * l + 1] & 0xFFFFFFFF) << 32)
                assert got == want, (l, "live" if arr is live else "need", bin(got), bin(want))
        return "lds"
    err = lib.mshgnn_hostplan_last_error().decode()
    assert err, "a refused descriptor carries a message"
    gmeta = (C.c_int32 * 16)()
    grc = lib.mshgnn_hostplan_compile_gen(C.byref(holder.desc), gmeta, 1)
    assert grc != -2, lib.mshgnn_hostplan_last_error().decode()
    if dtype == "f32":
        return "refused" if grc < 0 else "generic"
    assert grc > 0, f"neither engine takes it: {err} / {lib.mshgnn_hostplan_last_error().decode()}"
    assert gmeta[1] == layers and gmeta[2] == sum(n.values()) and gmeta[3] == hidden and gmeta[6] > 0 and gmeta[7] >= gmeta[6]
    return "generic"


@settings(max_examples=120, deadline=None, suppress_health_check=list(HealthCheck))
@given(case=topologies(), dtype=st.sampled_from(["bf16", "bf16", "x3", "f32"]))
def test_plan_tables_stay_in_bounds_and_liveness_matches(case, dtype):
    check_case(_lib(), *case, dtype)


def test_host_compilers_under_address_sanitizer():
    """The same walk under ASan + UBSan in a child process (the sanitizer runtime must be loaded first)."""
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan for this gcc")
    subprocess.run(["make", "-C", CSRC, "asan-host"], check=True, capture_output=True)
    code = ("import sys; sys.path.insert(0, %r); import random\n"
            "from tests import test_plan_property as t\n"
            "lib = t._lib(asan=True); rng = random.Random(7); taken = {}\n"
            "for i in range(60):\n"
            "    n = {'base': rng.randint(1, 4), 'joint': rng.randint(1, 40), 'foot': rng.randint(1, 40)}\n"
            "    rels = []\n"
            "    for s_, d_ in t.PAIRS:\n"
            "        k = rng.randint(1 if d_ == 'foot' else 0, min(48, 2 * max(n[s_], n[d_])))\n"
            "        rels.append(((s_, 'connect', d_), [[rng.randrange(n[s_]), rng.randrange(n[d_])] for _ in range(k)]))\n"
            "    e = t.check_case(lib, n, rels, rng.randint(1, 8), 128 * rng.choice([1, 1, 2, 4, 16]), rng.choice(['bf16', 'x3']))\n"
            "    taken[e] = taken.get(e, 0) + 1\n"
            "import bench\n"
            "for cfg, L in (('a1c2', 3), ('a1c2', 8), ('mck4', 8), ('solo', 8), ('synth32', 6)):\n"
            "    spec = bench.build_spec(L, cfg, 512 if cfg == 'synth32' else 128)\n"
            "    h = t.eng._DescHolder(spec, t.eng.DTYPE_CODES['bf16']); import ctypes as C\n"
            "    m = (C.c_int32 * 160)(); rc = lib.mshgnn_hostplan_compile(C.byref(h.desc), None, 0, m, 1)\n"
            "    assert rc != -2, lib.mshgnn_hostplan_last_error()\n"
            "    if rc < 0: assert lib.mshgnn_hostplan_compile_gen(C.byref(h.desc), (C.c_int32 * 16)(), 1) > 0, lib.mshgnn_hostplan_last_error()\n"
            "print('asan ok', taken)\n") % ROOT
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "asan ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_spec_tables_in_sync():
    """mshgnn_spec_tables.inc (the compile-time programs of the specialised step kernels) == what the plan compiler emits for those descriptors now.
    (A stale file is harmless at run time -- a plan whose tables differ keeps the interpreting kernel -- but it would silently lose the speed.)"""
    subprocess.run(["make", "-C", CSRC, "../libmshgnn_hostplan.so"], check=True, capture_output=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_spec_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_on_demand_program_tables_equal_the_generated_ones():
    """morphsym_hgnn_amd/jit.py renders a plan's tables the way tools/gen_spec_tables.py does: for a plan that HAS a built-in program the two texts hold the same rows."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    from morphsym_hgnn_amd import jit
    subprocess.run(["make", "-C", CSRC, "../libmshgnn_hostplan.so"], check=True, capture_output=True)
    spec = bench.build_spec(3, "a1c2")
    holder = eng._DescHolder(spec, eng.DTYPE_CODES["bf16"])
    tables, meta = jit.plan_tables(holder.desc)
    name, txt = jit.render_program(tables, meta, spec.out_channels)
    inc = open(os.path.join(CSRC, "mshgnn_spec_tables.inc")).read()
    built_in = inc[inc.index("struct A1C2_L3 {"):inc.index("struct A1C2_L8 {")]
    rows = lambda t: [l.strip() for l in t.splitlines() if l.strip().startswith("{")]
    assert name.startswith("JIT_") and rows(txt) == rows(built_in) and "PRE = 7" in txt and "PRE = 7" in built_in
