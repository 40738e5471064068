"""CPU tests: the C-ABI library loads here (no GPU) and exports exactly what include/glgym.h declares."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from gl_gym_amd import _lib
    if not _lib.LIB_PATH.exists():
        g.build()
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    header = (ROOT / "include" / "glgym.h").read_text()
    declared = set(re.findall(r"\b(glgym_[a-z_A-Z0-9]+)\s*\(", header))
    declared -= {"glgym_handle_s"}
    L = lib.load()
    assert declared == set(lib.PROTOTYPES), declared ^ set(lib.PROTOTYPES)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.glgym_version()


def test_struct_layouts_match_header_sizes(lib):
    # 4-byte ints followed by 8-byte pointers: ctypes applies the same natural alignment as the C compiler
    assert C.sizeof(lib.RewardCfg) == 16 * 8
    assert C.sizeof(lib.StepArgs) == 8 + 8 + 8 * 5 + 8 + 8 * 3 + 8 + 8 * 5        # struct_size, B | ld | ... + step_flags (ABI 5)
    assert lib.StepArgs._fields_[0][0] == "struct_size" and lib.make_step_args(1, 2).struct_size == C.sizeof(lib.StepArgs)
    hdr = (ROOT / "include" / "glgym.h").read_text()
    assert int(re.search(r"#define GLGYM_ABI_VERSION (\d+)", hdr).group(1)) == lib.ABI_VERSION == lib.load().glgym_abi_version()
    assert C.sizeof(lib.ObsArgs) == 8 + 8 * 3 + 8 + 8 * 3 + 8 + 8 + 8 + 8
    assert C.sizeof(lib.ResetArgs) == 8 + 8 * 5 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = lib.load()
    h = C.c_void_p()
    p = np.zeros(208)
    rc = L.glgym_create(28, 6, 10, 208, 900.0, p.ctypes.data_as(lib._DP), 0, 256, 0, C.byref(h))
    assert rc == lib.ENODEV and b"no CPU fallback" in L.glgym_last_error()
    rc = L.glgym_create(27, 6, 10, 208, 900.0, p.ctypes.data_as(lib._DP), 0, 256, 0, C.byref(h))
    assert rc == lib.EINVAL
    from gl_gym_amd import GreenLight, GlgymError
    with pytest.raises(GlgymError):
        GreenLight(28, 6, 10, 208, 900.0)
    from gl_gym_amd.tomato_env import TomatoVecEnv
    with pytest.raises(GlgymError):
        TomatoVecEnv(4)


def test_ode_pipe_defaults_to_the_scheme_its_kernels_are_built_for(lib):
    """the default scheme is "ls5", but GLGYM_ODE_PIPE is instantiated for GLGYM_SCHEME_RK4 only (glgym_step /
    glgym_evalF: GLGYM_EINVAL otherwise).  The constructors therefore resolve `scheme=None` per variant, and refuse an explicit
    other scheme by name instead of failing at the first step (the GPU half: tests/test_gpu_parity.py::test_ode_pipe_variant_and_nd14_rows)."""
    assert lib.resolve_scheme(None) == lib.DEFAULT_SCHEME == "ls5"
    assert lib.resolve_scheme(None, "ode_pipe") == "rk4" and lib.resolve_scheme("rk4", "ode_pipe") == "rk4"
    for sch in ("ls5", "rk3", "rk2"):
        assert lib.resolve_scheme(sch, "ode") == sch
        with pytest.raises(ValueError, match="ode_pipe"):
            lib.resolve_scheme(sch, "ode_pipe")
    with pytest.raises(ValueError):
        lib.resolve_scheme("euler")
    with pytest.raises(ValueError):
        lib.resolve_scheme(None, "ode_tube")
    # the presets follow the resolved scheme: rk4's parity count for the evalF class, its throughput count for the batched env
    assert lib.preset_n_sub("rk4", 300.0, "parity") == (216, 0) and lib.preset_n_sub("rk4", 300.0, "throughput") == (80, 0)


def test_product_never_imports_the_oracle():
    pkg = ROOT / "greenlight-gym2_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.hpp")):
        txt = f.read_text()
        assert "oracle" not in txt.replace("the oracle", "").replace("against the oracle", "") or f.name == "gl_model.hpp", f


def test_default_constant_image_is_current(tmp_path):
    """csrc/gl_default_const.inc (compile-time constants of the specialised kernels) matches the generator."""
    import subprocess
    from gl_gym_amd.parameters import init_default_params
    csrc = ROOT / "greenlight-gym2_amd" / "csrc"
    exe = tmp_path / "gen"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", f"-I{csrc}", "-o", str(exe),
                           str(csrc / "gen_default_const.cpp")])
    txt = " ".join(repr(float(v)) for v in init_default_params())
    out = subprocess.run([str(exe)], input=txt, capture_output=True, text=True, check=True).stdout
    assert out == (csrc / "gl_default_const.inc").read_text()


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """The device assembly of glgym.hip with the optimisation flags of csrc/Makefile (OPT), cross-compiled ONCE for the ISA tests
    below (no GPU needed; about a minute)."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists():
        pytest.skip("hipcc not available")
    csrc = ROOT / "greenlight-gym2_amd" / "csrc"
    out = tmp_path_factory.mktemp("isa") / "glgym.s"
    opt = (csrc / "Makefile").read_text().split("OPT =")[1].split("\n")[0].split()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + opt + ["--offload-arch=gfx950", "-std=c++17",
                           f"-I{ROOT / 'include'}", "-S", "--cuda-device-only", "-o", str(out), str(csrc / "glgym.hip")])
    return out.read_text()


def test_fp32_step_kernel_isa_has_no_mfma_no_scratch_no_spill_reloads(device_asm):
    """ISA regression: the hot fp32 kernels must stay MFMA-free, scratch-free and free of SGPR-spill
    reloads (v_readlane) in the specialised variant.  hipcc cross-compiles here without a GPU."""
    import re
    s = device_asm
    assert "v_mfma" not in s
    # <T, PER_ENV_CROP, DEFAULT_P, PIPE, SCH, OCC>: the one-wave-per-SIMD builds (OCC = 1) of the classical-RK4 kernels (SCH 0) and
    # the default-parameter Bogacki-Shampine kernel (SCH 2); the
    # OCC = 2 build is limited to 256 registers on purpose and spills to scratch
    # (round 5: plus the five-stage 2N scheme, SCH 3 -- the default -- with the default block, handle parameters and per-env crop blocks)
    for variant, allow_readlane in (("step_kernelIfLb0ELb1ELb0ELi0ELi1E", False), ("step_kernelIfLb0ELb0ELb0ELi0ELi1E", True),
                                    ("step_kernelIfLb1ELb1ELb0ELi0ELi1E", False), ("step_kernelIfLb0ELb1ELb0ELi2ELi1E", False),
                                    ("step_kernelIfLb0ELb1ELb0ELi3ELi1E", False), ("step_kernelIfLb0ELb0ELb0ELi3ELi1E", True),
                                    ("step_kernelIfLb1ELb1ELb0ELi3ELi1E", False)):
        m = re.search(r"^(_ZN\S*" + variant + r"\S*):", s, flags=re.M)
        body = s[m.start():]
        body = body[:body.index(".Lfunc_end")]
        desc = s[s.index(".amdhsa_kernel " + m.group(1)):]
        desc = desc[:desc.index(".end_amdhsa_kernel")]
        assert "buffer_store_dword" not in body, variant
        if allow_readlane:
            # parameters from the kernarg block: hipcc assembles the (capacity, capacity) register pairs of the packed balance
            # products by bouncing one 16-byte kernarg load through a stack slot -- once, in the prologue, never in the loops
            lines = body.split("\n")
            where = [k for k, line in enumerate(lines) if "scratch_" in line]
            first_loop = min(k for k, line in enumerate(lines) if "Parent Loop" in line)       # first nested loop = the integrator
            assert len(where) <= 8 and all(k < first_loop for k in where), (variant, where, first_loop)
            assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1)) <= 32, variant
        else:
            assert "scratch_" not in body, variant
            assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", desc), variant
        if not allow_readlane:
            # wave-uniform scalars of the attempt ladder (round 3: four attempts, verified mode) live in VGPR lanes around the
            # integrator; a handful, outside its sub-step loops
            assert body.count("v_readlane_b32") < 32, (variant, body.count("v_readlane_b32"))


def test_quad_kernel_isa_keeps_its_spill_free_inner_loops(device_asm):
    """ISA regression for the four-lanes-per-environment kernels, every shipped variant: the fp64 builds sit close to
    the 512-register limit and hipcc 7.2's spill code has produced wrong fp64 results on this kernel before.
    fp32: no scratch at all.  fp64 (coefficient blocks in LDS): at most 256 bytes of private segment reserved, none of it touched inside the sub-step
    loops (window-level bookkeeping at most; the shipped builds contain no scratch instruction at all), no out-of-line call, and the quad_perm DPP moves stay 32-bit -- gfx950 implements 64-bit
    DPP for row_newbcast only."""
    import re
    csrc = ROOT / "greenlight-gym2_amd" / "csrc"
    s = device_asm
    names = re.findall(r"^(_ZN\S*(?:step_kernel_quad|evalf_kernel_quad)I[fd]\S*):", s, flags=re.M)
    # every scheme (SCH 0 / 1 / 2 / 3) of both kernels in fp64, with and without per-env crop blocks (ODE_pipe is a run-time selection
    # inside them); fp32: four schemes x default / handle parameters
    # (round 5: + the two-rungs-at-a-time evalF kernels, four fp64 and four fp32, and the sequential fp32 evalF quad kernels)
    # (round 6: + the two-rungs-at-a-time STEP kernels for verified raw-control steps: four fp64, eight fp32 -- default / handle parameters)
    assert len([n for n in names if "quadId" in n]) == 24 and len([n for n in names if "quadIf" in n]) == 24, names
    # round 4: the only fp64 integrator on the device is this layout (no one-lane fp64 kernels, hence no LDS mailbox); fp64 builds
    # with the default block compiled in are not instantiated (0.7 % for six more kernels), and the Makefile must not bring back the
    # scheduler flag under which they -- and a separate ODE_pipe build -- came out wrong
    assert not re.search(r"^_ZN\S*(?:step_kernel|evalf_kernel)Id\S*:", s, flags=re.M)
    assert not [n for n in names if "step_kernel_quadIdLb1E" in n]
    assert "max-ilp" not in (csrc / "Makefile").read_text().split("OPT =")[1].split("\n")[0]
    for name in names:
        fp64 = "quadId" in name
        body = s[s.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        desc = s[s.index(".amdhsa_kernel " + name):]
        desc = desc[:desc.index(".end_amdhsa_kernel")]
        assert "v_mfma" not in body
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1)) <= (256 if fp64 else 0), name
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1)) <= 512, name
        assert not re.search(r"v_mov_b64_dpp|v_mov_b64.*quad_perm", body), name
        assert "s_swappc" not in body, name                   # everything inlined: a call at this register pressure spills the caller
        # no scratch access inside the innermost (sub-step) loops: walk the blocks, track the loop depth the compiler annotates
        depth = 0
        for line in body.split("\n"):
            d = re.search(r"Depth=(\d)", line)
            if d and ("in Loop" in line or "Loop Header" in line):
                depth = int(d.group(1))
            elif re.match(r"^\.LBB\d+_\d+:\s*$", line):
                depth = 0
            if "scratch_" in line:
                assert depth < 3, (name, line.strip())


def test_recorded_pmc_constants_belong_to_the_shipped_isa(device_asm):
    """bench.py prices its roofline block with rocprofv3 counters recorded once per variant (profiles/rNN_pmc_constants.json; it
    cannot collect counters itself).  Each record carries the static fingerprint of the kernel it was taken on -- total instructions,
    vector and packed instructions of the sub-step loop (tools/pk_share.py) -- and this test recompiles the device code: a kernel
    change without re-recording the counters (tools/profile_r05.sh) fails here instead of shipping a stale `roofline.frac`."""
    import json
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    from pk_share import loop_stats
    sys.path.insert(0, str(ROOT))
    import bench
    newest = bench.PMC_FILES[0]
    d = json.loads(newest.read_text())
    assert "f32_ls5" in d and "f64_ls5_quad" in d, sorted(d)          # the default workload and config 2 are priced
    for variant, rec in d.items():
        assert "isa_fragment" in rec, variant
        now = loop_stats(device_asm, rec["isa_fragment"])
        assert now is not None, (variant, rec["isa_fragment"])
        for k in ("isa_instr_total", "isa_valu_in_loop", "isa_pk_in_loop"):
            assert now[k] == rec[k], (variant, k, now[k], rec[k], "re-record the counters: tools/profile_r05.sh")
        assert bench.load_pmc(variant) is rec or bench.load_pmc(variant) == rec      # ... and it is this file bench.py reads


def test_scheme_table_and_default_sub_step_counts():
    """Host-side scheme constants agree with include/glgym.h, and the nominal sub-step keeps its length when dt changes
    (multiples of the scheme's tier-2b window: RK4 -> 4, three-stage scheme -> 3, midpoint -> 4)."""
    import re
    from gl_gym_amd import _lib as L
    hdr = (ROOT / "include" / "glgym.h").read_text()
    enum = dict((k, int(v)) for k, v in re.findall(r"(GLGYM_SCHEME_[A-Z0-9]+) = (\d)", hdr))
    assert enum == {"GLGYM_SCHEME_RK4": L.SCHEMES["rk4"], "GLGYM_SCHEME_RK2": L.SCHEMES["rk2"], "GLGYM_SCHEME_RK3": L.SCHEMES["rk3"],
                    "GLGYM_SCHEME_LS5": L.SCHEMES["ls5"]}
    assert L.DEFAULT_SCHEME == "ls5"
    assert [L.default_n_sub(s, 900.0) for s in ("ls5", "rk4", "rk3", "rk2")] == [128, 240, 270, 336]
    assert [L.default_n_sub(s, 300.0) for s in ("ls5", "rk4", "rk3", "rk2")] == [44, 80, 90, 112]
    assert [L.default_n_sub(s, 1800.0) for s in ("ls5", "rk4", "rk3", "rk2")] == [256, 480, 540, 672]
    assert L.default_n_sub("rk3", 1.0) == 3 and L.default_n_sub("rk4", 1.0) == 4 and L.default_n_sub("ls5", 1.0) == 2
    # the parity preset: ls5 with ONE sub-step per window; the others keep their window and take 8/3 of the nominal count
    assert L.preset_n_sub("ls5", 900.0, "parity") == (192, 1) and L.preset_n_sub("rk4", 900.0, "parity") == (640, 0)
    assert L.preset_n_sub("ls5", 300.0, "parity") == (64, 1) and L.preset_n_sub("ls5", 900.0, "throughput") == (128, 0)
    layouts = dict((k, int(v)) for k, v in re.findall(r"GLGYM_LAYOUT_([A-Z]+) = (\d)", hdr))
    assert layouts == {k.upper(): v for k, v in L.LAYOUTS.items()}
    assert int(re.search(r"#define GLGYM_METRIC_REPLICAS (\d+)", hdr).group(1)) == L.METRIC_REPLICAS
    assert int(re.search(r"#define GLGYM_METRIC_STRIDE (\d+)", hdr).group(1)) == L.METRIC_STRIDE >= L.NMETRIC
