"""CPU: the C-ABI library builds for gfx950, loads through ctypes without a GPU and exports exactly the entry
points include/dtk.h declares; the Python binding lists a signature for each; argument validation paths that do not
touch the device behave (error code + message)."""
import ctypes

import pytest

import __graft_entry__ as entry
from dino_tracker_amd import _lib


@pytest.fixture(scope="module")
def handle():
    entry.build()
    return _lib.lib()


def test_exports_match_header(handle):
    declared = entry.declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(handle, name), name
    assert sorted(_lib.SIGNATURES) == declared


def test_version_and_error_paths(handle):
    assert handle.dtk_version() == 1
    g = _lib.make_geom(4, 30, 140, 210)  # C=30 is not a multiple of 4
    rc = handle.dtk_sample_points(g, 1, 1, 1, None, 1, 1, None)
    assert rc == -1 and b"multiple of 4" in handle.dtk_last_error()
    g = _lib.make_geom(4, 32, 140, 210)
    g.ph += 1
    exact = _lib.TrackOpts(0, 0, 0, 0)
    rc = handle.dtk_track(g, 1, 1, None, 1, 1, None, 1, None, 1, 1, None, exact, None, 1, 1024, None)
    assert rc == -1 and b"inconsistent" in handle.dtk_last_error()
    g = _lib.make_geom(4, 32, 140, 210)
    rc = handle.dtk_track(g, 1, 1, None, 1, 1, None, 1, None, 1, 1, None, _lib.TrackOpts(0, 0, 0, 7), None, 1, 1024, None)
    assert rc == -1 and b"bad options" in handle.dtk_last_error()
    assert handle.dtk_track_workspace_bytes(g, 100, exact) == 100 * (576 + 1) * 4
    # the MFMA workspace follows the round size (smaller rounds -> smaller staging)
    big = handle.dtk_track_workspace_bytes(g, 100000, _lib.TrackOpts(1, 0, 0, 0))
    small = handle.dtk_track_workspace_bytes(g, 100000, _lib.TrackOpts(1, 0, 1024, 0))
    assert 0 < small < big


def test_production_library_has_no_development_switches(handle):
    """The shipped libdtk.so is built without -DDTK_DEV: no getenv("DTK_DEBUG"), no debug exports."""
    import subprocess
    for name in ("dtk_debug_counters", "dtk_debug_track_counts"):
        assert not hasattr(handle, name), name
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_hot_path_refuses_cpu_tensors():
    import torch
    from dino_tracker_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pack_features(torch.zeros(1, 4, 2, 2))


def test_struct_layouts_match_header():
    """The ctypes mirrors of the option / statistics structs have the size include/dtk.h's declarations imply (round 4 added
    dtk_track_opts.emb_rows), and the feat_f16 buffer grows by the split planes exactly at C = 384 and by the scale slot
    (no device needed)."""
    import re
    text = open(entry.os.path.join(entry.ROOT, "include", "dtk.h")).read()
    body = re.search(r"typedef struct dtk_track_opts \{(.*?)\} dtk_track_opts;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [f.strip() for decl in re.findall(r"int32_t ([^;]+);", body) for f in decl.split(",")]
    assert fields == [name for name, _ in _lib.TrackOpts._fields_]
    assert ctypes.sizeof(_lib.TrackOpts) == 4 * len(fields)
    handle = _lib.lib()
    for C, extra in ((384, True), (256, False), (1024, False)):
        g = _lib.make_geom(3, C, 140, 210)
        unit = 3 * g.ph * ((g.pw + 127) // 128 * 128) * C * 2
        total = handle.dtk_feat_f16_bytes(g)
        planes = 3 * g.ph * g.pw * C * 4
        body_bytes = (unit + 255) // 256 * 256 + planes if extra else unit
        # + the 256-byte slot of the window correlations' scale (round 6), behind the 256-byte-aligned body
        assert total == (body_bytes + 255) // 256 * 256 + 256, (C, total, unit, planes)
    assert handle.dtk_contrastive_workspace_bytes(16, 256, 384, 8107) > 16 * 256 * 8108 * 4 * 2


def test_adam_args_layout_matches_header():
    """dtk_adam_args travels by value: the ctypes mirror must have the header's fields in the header's order and its size."""
    import re
    text = open(entry.os.path.join(entry.ROOT, "include", "dtk.h")).read()
    body = re.search(r"typedef struct dtk_adam_args \{(.*?)\} dtk_adam_args;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [re.sub(r"\[.*", "", f.strip()).strip("* ") for decl in re.findall(r"(?:float\*|const float\*|int64_t|int32_t|float|double)\s+([^;]+);", body)
             for f in decl.split(",")]
    assert names == [n for n, _ in _lib.AdamArgs._fields_], names
    assert int(re.search(r"#define DTK_ADAM_MAX_TENSORS (\d+)", text).group(1)) == _lib.ADAM_MAX_TENSORS
    assert int(re.search(r"#define DTK_ADAM_MAX_GROUPS (\d+)", text).group(1)) == _lib.ADAM_MAX_GROUPS
    assert ctypes.sizeof(_lib.AdamArgs) == (4 * 8 + 8 + 4 + 4) * 32 + 4 + 4 + 4 * 8 + 3 * 8   # (4 bytes of padding in front of the doubles)


def test_m0_users(handle, tmp_path):
    """csrc/common.h dtk_buffer_lds16 (and its copy in vit_attention4.h) writes m0 inside an asm statement the compiler cannot be
    told about (hipcc refuses reserved registers in clobber lists), so the rule is structural: a kernel that contains the
    descriptor form of LDS-DMA (`s_add_u32 m0, ...` + `buffer_load_dwordx4 ... lds`) has NO other instruction that reads or writes
    m0.  Checked in the ISA of the built objects."""
    import os
    import re
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    csrc = os.path.dirname(_lib.LIB_PATH)
    seen = 0
    for name in ("vit", "track_mfma"):
        fat, co = str(tmp_path / f"{name}.fatbin"), str(tmp_path / f"{name}.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", os.path.join(csrc, f"{name}.o"), fat], check=True)
        subprocess.run([f"{llvm}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                        f"--output={co}", "--unbundle"], check=True)
        asm = subprocess.run([f"{llvm}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
        kern, users = None, {}
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                kern = m.group(1)
                continue
            ins = line.split("//")[0].strip()
            if re.search(r"\bm0\b", ins):
                users.setdefault(kern, []).append(ins)
        for kern, ins in users.items():
            dma = [i for i in ins if i.startswith("s_add_u32 m0,")]
            if dma:
                seen += 1
                assert len(dma) == len(ins), (kern, [i for i in ins if not i.startswith("s_add_u32 m0,")][:4])
    assert seen >= 4   # attention4 (x2 operand types), gemm_ws V2D forms, corr_peaks, refine_corr_dma


def test_vit_flag_values_match_header():
    """dtk_vit_model.flags: the Python constants (dino_tracker_amd/_lib.py) are the header's #defines (include/dtk.h)."""
    import re
    import os
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dtk.h")).read()
    want = {"DTK_VIT_TILED_GEMMS": _lib.VIT_TILED_GEMMS, "DTK_VIT_BF16": _lib.VIT_BF16, "DTK_VIT_CHECK_RANGE": _lib.VIT_CHECK_RANGE,
            "DTK_VIT_ATTENTION_V2": _lib.VIT_ATTENTION_V2, "DTK_VIT_GEMM_WS_V1": _lib.VIT_GEMM_WS_V1,
            "DTK_VIT_ATTENTION_V4": _lib.VIT_ATTENTION_V4, "DTK_VIT_GEMM_WIDE_V1": _lib.VIT_GEMM_WIDE_V1,
            "DTK_VIT_NO_LN_FUSION": _lib.VIT_NO_LN_FUSION}
    for name, value in want.items():
        m = re.search(rf"#define {name} (\d+)", text)
        assert m and int(m.group(1)) == value, name
    assert len(set(want.values())) == len(want) and all(v & (v - 1) == 0 for v in want.values())
