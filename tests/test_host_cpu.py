"""CPU-side tests of the product's host logic: the C-ABI library loads and exports every symbol the
header declares (no compute call without a GPU), geometry helpers, weight packing, config checks,
container plumbing, and the loud failure when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_state_dict


def test_library_exports_every_header_symbol():
    from llicti_amd import _lib
    _lib.build()
    hdr = open(os.path.join(ROOT, "include", "llicti_hip.h")).read()
    declared = set(re.findall(r"\b(llicti_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("llicti_ctx")
    L = C.CDLL(_lib.SO_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(_lib.EXPORTS)
    assert b"gfx950" in _lib.lib().llicti_version()


def test_header_is_plain_c_and_native_client_builds(tmp_path):
    """include/llicti_hip.h is the boundary a non-Python host binds: it must compile as plain C (gcc, no HIP, no C++) -- plain pointers and sizes,
    no torch type in any signature -- and tools/native_client.cpp (the C-ABI from a host with no Python in the process; run on the GPU box by
    tests/test_hip_parity.py::test_native_client_without_torch) must build against it and link with the in-tree library (hipcc cross-compiles)."""
    import shutil
    import subprocess
    from llicti_amd import _lib
    _lib.build()
    hdr = os.path.join(ROOT, "include", "llicti_hip.h")
    csrc = tmp_path / "use_header.c"
    csrc.write_text('#include "llicti_hip.h"\nint main(void) { llicti_ctx *c = 0; return llicti_workspace_bytes(1, 64, 64, 0) == 0 && c == 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.dirname(hdr), str(csrc)])
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc here")
    so_dir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.dirname(hdr), os.path.join(ROOT, "tools", "native_client.cpp"),
                           "-L", so_dir, "-lllicti_hip", "-Wl,-rpath," + so_dir, "-o", str(tmp_path / "native_client")])
    assert (tmp_path / "native_client").exists()


def test_magic_division_selftest():
    """The stage geometry's division by an invariant width (div_magic / div_by_magic, used by every decoder to turn a symbol
    index into a row and a column) against '/': every divisor 1 .. 8192, boundary and pseudo-random dividends below 2^31."""
    from llicti_amd import _lib
    assert _lib.lib().llicti_selftest() == 0, _lib.lib().llicti_last_error()


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from llicti_amd import _lib
    from llicti_amd.codec import HipCodec
    ctx = C.c_void_p()
    rc = _lib.lib().llicti_create(C.byref(ctx), 0)
    assert rc == _lib.ENODEVICE and b"no CPU path" in _lib.lib().llicti_last_error()
    with pytest.raises(_lib.LlictiError):
        HipCodec()
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    with pytest.raises(_lib.LlictiError):
        LLICTI(default_config()).compress(torch.zeros(1, 3, 32, 32))


@pytest.mark.parametrize("H,W", [(32, 32), (67, 93), (512, 768), (2160, 3840), (33, 250)])
def test_level_geometry_matches_oracle(H, W):
    from llicti_amd import _lib
    from oracle import oracle as orc
    for lvl in range(5):
        for band in range(3):
            Hl, Wl, h, w, padH, padW, hc, wc = _lib.level_geom(H, W, lvl, band)
            assert (Hl, Wl, h, w, padH, padW) == orc.level_geom(H, W, lvl)
            assert hc == (h - padH if band in (0, 2) else h) and wc == (w - padW if band in (0, 1) else w)
    # position / symbol counts
    if (H, W) == (512, 768):
        assert sum(_lib.level_geom(H, W, l)[2] * _lib.level_geom(H, W, l)[3] for l in range(5)) == 130944
    if (H, W) == (2160, 3840):
        assert _lib.level_geom(H, W, 4)[2:6] == (68, 120, 1, 0)


def test_workspace_and_container_bounds():
    from llicti_amd import _lib
    L = _lib.lib()
    assert L.llicti_workspace_bytes(1, 16, 16, 0) == 0          # too small
    assert L.llicti_workspace_bytes(1, 9000, 64, 0) == 0        # header stores h4 as uint8
    n1 = L.llicti_workspace_bytes(1, 512, 768, 0)
    n24 = L.llicti_workspace_bytes(24, 512, 768, 0)
    assert 0 < n1 < n24 < 8 * 2 ** 30
    # worst case of a 16-bit-precision coder: 2 bytes per symbol + termination / stream flush, 1,178,496 symbols
    assert 2 * 1178496 < L.llicti_max_container_bytes(512, 768) < 2 * 1178496 + 256 * 1024
    assert L.llicti_workspace_bytes(1, 512, 768, 0x100 | 8) > 0
    assert L.llicti_workspace_bytes(1, 512, 768, 0x100 | 3) > 0 and L.llicti_workspace_bytes(1, 512, 768, 0x100 | 128) > 0
    assert L.llicti_workspace_bytes(1, 512, 768, 0x100 | 33) == 0    # M in 1 .. 32, 64 or 128
    assert L.llicti_workspace_bytes(1, 512, 768, 0x100 | 96) == 0
    assert L.llicti_workspace_bytes(1, 512, 768, 7) == 0
    # wide streams (128 lanes): M in 1 .. 30
    assert L.llicti_workspace_bytes(24, 512, 768, 0x300 | 10) > L.llicti_workspace_bytes(24, 512, 768, 0x100 | 10)
    assert L.llicti_workspace_bytes(1, 512, 768, 0x300 | 14) > 0
    assert L.llicti_workspace_bytes(1, 512, 768, 0x300 | 15) == 0 and L.llicti_workspace_bytes(1, 512, 768, 0x300) == 0
    # xwide streams (256 lanes, v4): 1 .. 32, 64, 128
    assert L.llicti_workspace_bytes(24, 512, 768, 0x500 | 9) > L.llicti_workspace_bytes(24, 512, 768, 0x300 | 9)
    assert L.llicti_workspace_bytes(1, 512, 768, 0x500 | 64) > 0 and L.llicti_workspace_bytes(1, 512, 768, 0x500 | 32) > 0
    assert L.llicti_workspace_bytes(1, 512, 768, 0x500 | 15) > 0 and L.llicti_workspace_bytes(1, 512, 768, 0x500 | 128) > 0
    assert L.llicti_workspace_bytes(1, 512, 768, 0x500 | 33) == 0 and L.llicti_workspace_bytes(1, 512, 768, 0x500) == 0 and L.llicti_workspace_bytes(1, 512, 768, 0x700 | 4) == 0
    assert L.llicti_workspace_bytes(1, 512, 768, 0x200 | 4) == 0


def test_weight_packing_and_state_dict_names():
    import torch
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    from llicti_amd.weights import expected_keys, pack_state_dict
    torch.manual_seed(1337)
    m = LLICTI(default_config())
    sd = m.state_dict()
    ref = load_state_dict("rand1337")       # captured from the reference with the same seed (base.py:28)
    assert set(sd) == set(ref) and len(sd) == 33
    assert all(np.array_equal(sd[k].numpy(), ref[k]) for k in ref)
    assert sum(p.numel() for p in m.parameters()) == 196596      # exp_debug.log:101 (0.750 MB)
    assert set(expected_keys()) <= set(sd)
    packed = pack_state_dict(sd)
    assert [packed[b]["K0"] for b in range(3)] == [48, 72, 120]
    assert packed[2]["w0"].shape == (352, 120)
    w = ref["entropymodel.entmdls_scale_band.0.2.layer0_11_10.weight"]
    assert np.array_equal(packed[2]["w0"][:, 36:72], w.reshape(352, -1))
    # a real checkpoint carries extra compressai buffers: they must be ignorable
    sd2 = dict(ref)
    sd2["entropymodel.entmdls_scale_band.0.0.conditional_prob_model._offset"] = np.zeros(1)
    assert np.array_equal(pack_state_dict(sd2)[0]["w0"], packed[0]["w0"])


def test_config_rejects_unsupported_variants():
    from llicti_amd.config import check_supported, default_config
    check_supported(default_config())
    for k, v in [("num_mixtures", 3), ("clr_joint_mode", 0), ("ycocg", False), ("activfun", "LeakyReLU"),
                 ("dwtlevels", [0, 1, 2, 3]), ("mwsa_joint", True)]:
        with pytest.raises(NotImplementedError):
            check_supported(default_config(**{k: v}))


def test_container_list_roundtrip():
    from llicti_amd.codec import bytestream_list_to_container, container_to_bytestream_list, header_dims
    rng = np.random.default_rng(0)
    segs = [bytes([5, 3, 4]), bytes(12), bytes([0x6A, 0x03]), bytes(rng.integers(0, 256, 36, dtype=np.uint8))]
    segs += [bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) for n in rng.integers(0, 40, 45)]
    buf = np.frombuffer(b"".join(segs), np.uint8)
    seg_len = np.array([len(s) for s in segs], np.int32)
    bl = container_to_bytestream_list(buf, seg_len)
    assert len(bl) == 6 and all(len(r) == 9 for r in bl) and bl[0][4:] == [b""] * 5
    b2, s2 = bytestream_list_to_container(bl)
    assert np.array_equal(b2, buf) and np.array_equal(s2, seg_len)
    with pytest.raises(ValueError):
        bytestream_list_to_container(bl[:5])
    # header -> image size: h4=3, w4=4, pad flags 0x036A = 874 is the fixture value of a 67x93 image
    assert header_dims(bytes([5, 3, 3]) + bytes(12) + bytes([0x6A, 0x03])) == (67, 93)
    assert header_dims(bytes([5, 16, 24]) + bytes(12) + bytes([0, 0])) == (512, 768)


def test_rate_logger_matches_reference_text():
    """loggers/rate.py:120-168: the table text the reference's own RateLogger printed for this input
    (tests/golden/rate_table.json, generated by tests/golden/make_fixtures_f.py from the reference module)."""
    import json
    import logging
    from llicti_amd.agents.llicti_agent import CompressionRLossList
    from llicti_amd.loggers.rate import RateLogger
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "rate_table.json")))
    bl = [[bytes(n) for n in row] for row in g["stream_lengths"]]
    rates = CompressionRLossList().forward(g["numel"], bl)               # graphs/losses/rate_dist.py:130-135
    assert np.allclose(rates, g["rates"], rtol=0, atol=0)
    lines = []

    class Grab(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    lg = logging.getLogger("Rate Loss")
    h = Grab()
    lg.addHandler(h)
    lg.setLevel(logging.INFO)
    try:
        rl = RateLogger()
        rl._get_time_now_str = lambda: "12:34:56"
        rl(rates)
        rl(g["rates2"])
        tot, zero = rl.display(typ="te")
        assert lines[-1] == g["text"]["te"]
        assert abs(float(tot) - g["display_sum"]) < 1e-12 and zero == 0.0
        rl(rates)
        rl.display(lr=0.0001, typ="va")
        assert lines[-1] == g["text"]["va"]
        with pytest.raises(AssertionError):
            rl([[0.0] * 8] * 6)
            rl.display(typ="te")
    finally:
        lg.removeHandler(h)


def test_llic_file_and_ppm_roundtrip(tmp_path):
    from llicti_amd import fileio
    rng = np.random.default_rng(0)
    bl = [[bytes([5, 2, 3]), rng.integers(0, 256, 12, dtype=np.uint8).tobytes(), b"\x00\x00", rng.integers(0, 256, 18, dtype=np.uint8).tobytes()] + [b""] * 5]
    bl += [[rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in rng.integers(0, 300, 9)] for _ in range(5)]
    p = tmp_path / "a.llic"
    fileio.write_llic(p, bl)
    assert fileio.read_llic(p) == bl
    raw = p.read_bytes()
    assert raw[:4] == b"LLIC" and raw[5] == 49 and len(raw) == 6 + 4 * 49 + sum(len(s) for r in bl for s in r)
    for bad in (raw[:-1], b"XLIC" + raw[4:], raw[:4] + b"\x09" + raw[5:], raw + b"\0"):
        with pytest.raises(ValueError):
            fileio.loads_llic(bad)
    with pytest.raises(ValueError):
        fileio.dumps_llic(bl[:5])
    rgb = rng.integers(0, 256, (3, 37, 51), dtype=np.uint8)
    for ext in (".ppm", ".png"):
        q = tmp_path / ("img" + ext)
        fileio.write_image(q, rgb)
        assert np.array_equal(fileio.read_image(q), rgb)
    (tmp_path / "c.ppm").write_bytes(b"P6\n# a comment\n51 37\n255\n" + rgb.transpose(1, 2, 0).tobytes())
    assert np.array_equal(fileio.read_image(tmp_path / "c.ppm"), rgb)


def test_div255_shortcut_is_exact():
    """numerics.hpp::div255_exact -- q = x * RN(1/255); r = fma(-q, 255, x); q' = fma(r, RN(1/255), q) -- equals the
    correctly rounded x / 255 for every half-integer |x| <= 350 (all regular sample points of any table), so the
    kernels' three-operation form and the oracle's IEEE division define the same grid.  Exact rational arithmetic."""
    from fractions import Fraction as Fr

    def rn(fr):                                    # round a rational to the nearest float32, ties to even
        c = np.float32(float(fr))
        cand = [c, np.nextafter(c, np.float32(np.inf)), np.nextafter(c, np.float32(-np.inf))]
        return min(cand, key=lambda v: (abs(Fr(float(v)) - fr), int(np.float32(v).view(np.uint32)) & 1))
    r255 = rn(Fr(1, 255))
    assert float(r255).hex() == "0x1.0101020000000p-8"          # the constant in numerics.hpp
    for k in range(-700, 701):
        x = Fr(k, 2)
        q = rn(x * Fr(float(r255)))
        rem = rn(x - Fr(float(q)) * 255)
        q2 = rn(Fr(float(q)) + Fr(float(rem)) * Fr(float(r255)))
        assert q2 == rn(x / 255), k


def test_model_size_line_matches_reference_log():
    """agents/llicti_agent.py:167-192 prints the one reference-logged number that is reproducible exactly:
    " model param+buffer=total size: 0.750+0.000=0.750MB" (experiments/.../exp_debug.log:101)."""
    import logging
    import torch
    from llicti_amd.agents.llicti_agent import LLICTIAgent
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    a = LLICTIAgent.__new__(LLICTIAgent)              # the constructor needs a GPU; the size estimate does not
    a.logger = logging.getLogger("Agent")
    a.model = LLICTI(default_config())
    n_par, n_buf = a.model_size_estimation()
    assert n_par == 196596 * 4 and n_buf == 9 * 4
    assert a.size_text == " model param+buffer=total size: 0.750+0.000=0.750MB"


def test_strict_checkpoint_loading():
    """compressai's known extra buffers are dropped, anything else that does not match raises."""
    import pytest
    import torch
    from llicti_amd.config import default_config
    from llicti_amd.graphs.models.LLICTI_nets import LLICTI
    from llicti_amd.weights import load_reference_state_dict
    torch.manual_seed(5)
    src = LLICTI(default_config())
    sd = dict(src.state_dict())
    p = "entropymodel.entmdls_scale_band.0.1.conditional_prob_model."
    sd[p + "_offset"] = torch.zeros(3, dtype=torch.int32)
    sd[p + "scale_table"] = torch.zeros(64)
    torch.manual_seed(6)
    dst = LLICTI(default_config())
    load_reference_state_dict(dst, sd)
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    with pytest.raises(KeyError):
        load_reference_state_dict(dst, {"module." + k: v for k, v in sd.items()})
    short = dict(sd)
    short.pop("entropymodel.entmdls_scale_band.0.2.layers1toL.2.bias")
    with pytest.raises(KeyError):
        load_reference_state_dict(dst, short)
    extra = dict(sd)
    extra["entropymodel.something_else.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        load_reference_state_dict(dst, extra)


def test_oracle_is_test_infrastructure_only():
    """The CPU oracle is the checker, never the product: no module of the package and no tool imports, loads or runs
    anything under oracle/ (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may), and the package
    has no CPU fallback to fall into -- its library loader raises without the HIP .so."""
    import os
    import re
    from conftest import ROOT
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|^\s*#\s*include\s*[<\"][^>\"]*oracle|liboracle|oracle/_ref/", re.M)
    bad = []
    for top in ("llicti_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".sh", ".hip", ".hpp", ".h")):
                    p = os.path.join(dirpath, f)
                    if pat.search(open(p, errors="replace").read()):
                        bad.append(os.path.relpath(p, ROOT))
    assert bad == [], bad
    src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import|import oracle", src)]
    assert uses, "bench.py's cpu_baseline leg times the oracle"
    for u in uses:                                     # every import sits inside a function (the baseline / bit-exactness legs), not at module level
        line_start = src.rfind("\n", 0, u) + 1
        assert src[line_start:u].strip() == "" and u - line_start >= 4, src[line_start:u + 40]


def test_dropin_import_aliases():
    """the reference's own import lines (main.py:4, :29-33; agents/llicti_agent.py:1-12) resolve to this package after
    llicti_amd.dropin.install().  Run in a child process: the aliases live in sys.modules."""
    import subprocess
    import sys
    code = (
        "import llicti_amd.dropin as d; d.install(); d.install()\n"          # idempotent
        "from agents import *\n"
        "from graphs.models.LLICTI_nets import LLICTI\n"
        "from graphs.losses.rate_dist import CompressionRLossList, TrainRLossList\n"
        "from loggers.rate import RateLogger\n"
        "import llicti_amd.graphs.models.LLICTI_nets as m, llicti_amd.agents.llicti_agent as a\n"
        "assert LLICTI is m.LLICTI and globals()['LLICTIAgent'] is a.LLICTIAgent\n"      # main.py:30 looks the class up in globals()
        "assert CompressionRLossList().forward(3 * 4 * 4, [[b'ab', b''], [b'c']]) == [[1.0, 0.0], [0.5]]\n"
        "import types, sys\n"
        "d.uninstall(); sys.modules['graphs'] = types.ModuleType('graphs')\n"   # a foreign 'graphs' already imported: refuse to mix
        "try:\n    d.install(); raise SystemExit('mixed import accepted')\nexcept ImportError:\n    pass\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_synth_images_are_the_fixture_generators():
    """llicti_amd.synth (what bench.py draws) == tests/helpers.make_image == tests/golden/make_fixtures.make_image: a fixture names an
    image by (kind, H, W, seed)."""
    from helpers import make_image
    from llicti_amd import synth
    for kind, H, W, seed in (("noise", 33, 47, 3), ("smooth", 40, 64, 11)):
        assert np.array_equal(synth.make_image(kind, H, W, seed), make_image(kind, H, W, seed))
    with pytest.raises(ValueError):
        synth.make_image("photo", 32, 32, 0)


def test_host_plan_logic_driver(tmp_path):
    """The library's HOST logic (llicti_amd/csrc/host_types.hpp, cnn_pack.hpp, host_plan.hpp: plan building for equal and mixed sizes,
    container tags, size bounds, header parsing, the CNN weight pack) has no HIP dependency: g++ compiles tests/sanitize_host.cpp against it
    and the driver checks every plan's internal consistency over the shapes and modes the suite uses.  (tests/sanitize_host.sh runs the same
    driver under AddressSanitizer + UBSan; here it is built plain, so that the CPU suite needs no sanitizer runtime.)"""
    import subprocess
    exe = tmp_path / "sanitize_host"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tests", "sanitize_host.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "clean" in out.stdout


def test_no_scratch_in_mfma_and_stage_kernels():
    """no kernel of the library may spill a vector register or use a private segment -- the band CNN (18 instantiations: band x
    tile rows x equal / mixed sizes) and the stage decoders least of all.  Round 5's `band_params_kernel<2,16,true>` (the heaviest kernel of the
    mixed-size path) carried 8 spilled VGPRs at the 128-register cap of a 1024-thread workgroup, and `<2,4,false>` a 176-byte stack frame: its
    staging lambda was a real call.  hipcc's own metadata (`-S --cuda-device-only`, no GPU needed) is the witness: tools/kernel_resources.py."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from kernel_resources import kernel_resources
    finally:
        sys.path.pop(0)
    rows = kernel_resources()
    names = [r["demangled"] for r in rows]
    cnn = [r for r in rows if "band_params_kernel" in r["name"]]
    stage = [r for r in rows if "rans_decode_stage" in r["name"] or "rans_tail_kernel" in r["name"] or "rans_encode_kernel" in r["name"]]
    assert len(cnn) == 18, names                        # 3 bands x {16, 8, 4} tile rows x {equal, mixed} sizes
    assert len(stage) >= 9, names                       # three stage decoders + encoder / tail per lane kind
    for r in rows:
        assert r["vgpr_spill_count"] == 0, (r["demangled"], r)
        assert r["private_segment_fixed_size"] == 0 and not r["uses_dynamic_stack"], (r["demangled"], r)
    for r in cnn:                                       # a 1024-thread workgroup has 128 unified registers per lane, a 512-thread one 256
        cap = 512 * 256 // r["max_flat_workgroup_size"]
        assert r["vgpr_count"] + r["agpr_count"] <= cap, (r["demangled"], r)


def test_header_mode_c_and_python_agree_on_every_tag():
    """llicti_header_mode (the C-ABI's reading of a header, what a native caller hands to llicti_decode_images*) against llicti_amd.codec.mode_of_header
    over EVERY byte 0 and every value of the pad field's stream-count bits: the same mode, or both refuse (retired v2 / xwide-v3 tags, a count in a
    container that is not xwide v4, unknown tags).  Host code only -- no GPU."""
    from llicti_amd import _lib
    from llicti_amd.codec import MODE_AC, MODE_RANS, mode_of_header, rans_pad_hi, rans_tag
    L = _lib.lib()
    accepted = set()
    for b0 in range(256):
        for u in range(64):
            hdr = bytes([b0, 3, 3]) + bytes(12) + int(0x036A | (u << 10)).to_bytes(2, "little")       # a 67x93 image's pad flags under the count
            m = C.c_int(-12345)
            rc = L.llicti_header_mode((C.c_uint8 * 17).from_buffer_copy(hdr), C.byref(m))
            try:
                want = mode_of_header(hdr)
            except ValueError:
                want = None
            if want is None:
                assert rc == _lib.EFORMAT, (hex(b0), u, rc, m.value)
            else:
                assert rc == 0 and m.value == want, (hex(b0), u, rc, m.value, want)
                accepted.add(want)
    # what is accepted is exactly what the encoder can write: the reference format, 1 .. 32 / 64 / 128 narrow, 1 .. 14 wide, 1 .. 32 / 64 / 128 xwide streams
    counts = list(range(1, 33)) + [64, 128]
    assert accepted == {MODE_AC} | {MODE_RANS(M) for M in counts} | {MODE_RANS(M, wide=1) for M in range(1, 15)} | {MODE_RANS(M, wide=2) for M in counts}
    for M in counts:
        hdr = bytes([rans_tag(M, 2), 3, 3]) + bytes(12) + int(0x036A | (rans_pad_hi(M, 2) << 10)).to_bytes(2, "little")
        assert mode_of_header(hdr) == MODE_RANS(M, wide=2)
