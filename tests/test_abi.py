"""CPU-side checks of the C ABI: the library loads, exports every symbol include/mgacbam.h declares, its pure-host entry
points (layouts, sizes, argument validation) behave, and the ctypes mirror of the structs matches the header.
No kernel is launched here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mgacbam.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mga(?:cbam|seg|pmg|kendall|head)_\w+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    fns = declared_functions()
    for must in ("mgacbam_forward", "mgacbam_backward", "mgacbam_forward_stages", "mgacbam_backward_stages",
                 "mgacbam_ctx_bytes", "mgacbam_bwd_scratch_bytes", "mgacbam_ctx_layout", "mgacbam_last_error",
                 "mgacbam_abi_version", "mgacbam_build_info", "mgacbam_resize_nearest"):
        assert must in fns


def test_library_exports_every_declared_symbol(built_lib):
    lib = C.CDLL(built_lib)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/mgacbam.h but not exported"


def test_binding_covers_every_declared_symbol(built_lib):
    from mga_yolo_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared_functions()
    lib = _lib.load()
    assert lib.mgacbam_abi_version() == _lib.ABI_VERSION
    hdr = int(re.search(r"#define MGACBAM_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert hdr == _lib.ABI_VERSION
    assert b"gfx950" in lib.mgacbam_build_info()


def test_stage_masks_match_header(built_lib):
    from mga_yolo_amd import _lib
    src = open(HEADER).read()
    enum = {k: int(v) for k, v in re.findall(r"(MGACBAM_\w+) = (\d+)", src)}
    assert enum["MGACBAM_FWD_FUSE"] == _lib.FWD_FUSE
    assert enum["MGACBAM_FWD_ALL"] == _lib.FWD_ALL and enum["MGACBAM_BWD_ALL"] == _lib.BWD_ALL
    assert enum["MGACBAM_BWD_PARAMS"] == _lib.BWD_PARAMS and enum["MGACBAM_BWD_INPUTS"] == _lib.BWD_INPUTS
    assert enum["MGACBAM_BWD_FUSE"] == _lib.BWD_FUSE
    for k, v in _lib.FWD_STAGES.items():
        assert enum["MGACBAM_FWD_" + k.upper()] == v
    names = dict(reduce1="REDUCE1", convT="CONVT", reduce2="REDUCE2", wsa="WSA", params="PARAMGRAD", apply="APPLY")
    for k, v in _lib.BWD_STAGES.items():
        assert enum["MGACBAM_BWD_" + names[k]] == v
    assert sum(_lib.FWD_STAGES.values()) == _lib.FWD_ALL
    assert sum(_lib.BWD_STAGES.values()) | _lib.BWD_FUSE == _lib.BWD_ALL


@pytest.mark.parametrize("shape", [(32, 64, 80, 80, 4), (2, 256, 20, 20, 16), (1, 48, 17, 17, 3), (8, 512, 80, 80, 32), (1, 8, 9, 7, 1)])
def test_ctx_layout_is_packed_aligned_and_ordered(built_lib, shape):
    from mga_yolo_amd import _lib
    B, Cc, H, W, hid = shape
    lay = _lib.ctx_layout(B, Cc, H, W, hid)
    assert lay["total"] == _lib.ctx_bytes(B, Cc, H, W, hid)
    order = [n for n in _lib.CTX_FIELDS if n not in ("total", "status")]
    assert lay["status"] == lay["sync"] + 4 * B * ((H * W + 15) // 16 + 1)      # the time-out word follows the k_gate tile flags
    sizes = dict(S=B, use=B, den=B, avg=B * Cc, mx=B * Cc, mavg=B * Cc, valid=B * Cc, amax=B * Cc, h_avg=B * hid, h_mx=B * hid,
                 ca=B * Cc, planes=B * 3 * H * W, cidx=B * H * W, sa=B * H * W,
                 proj=B * hid * H * W if hid <= _lib.PROJ_MAX_HIDDEN else 0, sync=6 * B * ((H * W + 15) // 16 + 1) + 4 + B + B * Cc)
    prev_end = 0
    for n in order:
        assert lay[n] % 16 == 0 and lay[n] >= prev_end, n
        prev_end = lay[n] + 4 * sizes[n]
    assert lay["total"] >= prev_end
    assert _lib.scratch_bytes(B, Cc, H, W, hid, 7) > 0


def test_argument_errors_are_reported_not_launched(built_lib):
    from mga_yolo_amd import _lib
    lib = _lib.load()
    assert lib.mgacbam_ctx_bytes(0, 64, 8, 8, 4) == 0 and b"bad shape" in lib.mgacbam_last_error()
    assert lib.mgacbam_bwd_scratch_bytes(1, 8, 8, 8, 1, 4) == 0 and b"odd" in lib.mgacbam_last_error()   # even k
    # mask head: rows wider than a pixel run holds are a shape error from the size queries on (the module runs torch ops for them)
    assert lib.mgahead_ctx_bytes(1, 8, 2, 500, 8) > 0
    assert lib.mgahead_ctx_bytes(1, 8, 2, 501, 8) == 0 and b"W <= 500" in lib.mgacbam_last_error()
    assert lib.mgahead_bwd_scratch_bytes(1, 8, 1 << 14, 1 << 14, 8) == 0        # hidden * H * W beyond a buffer descriptor
    lv = (_lib.FwdLevel * 1)()
    assert lib.mgacbam_forward(lv, 1, None) == -1            # MGACBAM_E_NULL: x / y / ctx missing
    assert lib.mgacbam_forward(lv, 0, None) == -5            # MGACBAM_E_LEVELS
    assert lib.mgacbam_forward(None, 1, None) == -1
    bl = (_lib.BwdLevel * 1)()
    assert lib.mgacbam_backward(bl, 1, None) == -1
    assert lib.mgacbam_resize_nearest(None, None, 1, 4, 4, 2, 2, None) == -1
    with pytest.raises(RuntimeError, match="argument error"):
        _lib.check(-2, "x")


STRUCTS = {          # header struct tag -> ctypes mirror (every struct include/mgacbam.h declares)
    "mgacbam_params": "Params", "mgacbam_fwd_level": "FwdLevel", "mgacbam_bwd_level": "BwdLevel", "mgacbam_ctx_layout": "CtxLayout",
    "mgacbam_eca_params": "EcaParams", "mgacbam_eca_fwd_level": "EcaFwdLevel", "mgacbam_eca_bwd_level": "EcaBwdLevel",
    "mgaseg_level": "SegLevel", "mgaseg_cfg": "SegCfg", "mgahead_params": "HeadParams", "mgahead_fwd_level": "HeadFwdLevel",
    "mgahead_bwd_level": "HeadBwdLevel", "mgapmg_cfg": "PmgCfg",
}


def _header_fields(struct):
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\}" % struct, src, re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(",")
        out.append(names[0].split()[-1].lstrip("*"))
        out += [n.strip().lstrip("*") for n in names[1:]]
    return out


def test_every_struct_of_the_header_has_a_mirror():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    assert sorted(re.findall(r"typedef struct (\w+) \{", src)) == sorted(STRUCTS)


@pytest.mark.parametrize("struct", sorted(STRUCTS))
def test_struct_mirrors_have_the_header_field_order(struct):
    from mga_yolo_amd import _lib
    assert _header_fields(struct) == [f[0] for f in getattr(_lib, STRUCTS[struct])._fields_]


def test_struct_mirrors_have_the_compiler_s_offsets_and_sizes(tmp_path):
    """The C compiler's own layout of every struct (sizeof + offsetof of each field, from a program that includes the header)
    against the ctypes mirrors: field order alone does not see a wrong width or a missed padding."""
    import shutil
    import subprocess
    from mga_yolo_amd import _lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mgacbam.h"', "int main(void) {"]
    for tag in sorted(STRUCTS):
        lines.append(f'  printf("{tag} size %zu\\n", sizeof({tag}_t));')
        for f in _header_fields(tag):
            lines.append(f'  printf("{tag} {f} %zu\\n", offsetof({tag}_t, {f}));')
    lines += ["  return 0;", "}"]
    (tmp_path / "layout.c").write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.run([cc, "-I", os.path.dirname(HEADER), str(tmp_path / "layout.c"), "-o", exe], check=True)
    got = {}
    for ln in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines():
        tag, field, val = ln.split()
        got[(tag, field)] = int(val)
    for tag, mirror in STRUCTS.items():
        cls = getattr(_lib, mirror)
        assert got[(tag, "size")] == C.sizeof(cls), tag
        for name, _ in cls._fields_:
            assert got[(tag, name)] == getattr(cls, name).offset, (tag, name)


def _fake(addr=0x10000):
    return addr          # a non-NULL, 16-byte aligned "pointer": the calls below must fail before anything dereferences it


def test_undersized_work_buffers_are_an_error_not_a_launch(built_lib):
    """ABI 14: every work buffer travels with its capacity and every entry point checks it against the CURRENT requirement before
    launching: MGACBAM_E_SIZE, last_error names the buffer with want / got.  (No GPU here: a call that got past the check would fail
    differently.)"""
    from mga_yolo_amd import _lib
    lib = _lib.load()
    B, Cc, H, W, hid, k = 2, 64, 16, 16, 4, 7
    P = _lib.Params(*([_fake()] * 6), hid, k, 1, 1e-4, 1e-6)
    need_ctx, need_scr = _lib.ctx_bytes(B, Cc, H, W, hid), _lib.scratch_bytes(B, Cc, H, W, hid, k)
    fl = (_lib.FwdLevel * 1)()
    F = fl[0]
    F.x = F.mask = F.y = F.ctx = _fake()
    F.p, F.B, F.C, F.H, F.W, F.dtype = P, B, Cc, H, W, _lib.F32
    F.ctx_bytes = need_ctx - 16
    assert lib.mgacbam_forward(fl, 1, None) == _lib.E_SIZE
    msg = lib.mgacbam_last_error().decode()
    assert "ctx" in msg and str(need_ctx) in msg and str(need_ctx - 16) in msg
    with pytest.raises(RuntimeError, match="too small"):
        _lib.check(_lib.E_SIZE, "x")
    bl = (_lib.BwdLevel * 1)()
    Bw = bl[0]
    for f in ("x", "mask", "gy", "ctx", "scratch", "gx", "gmask", "gw1", "gb1", "gw2", "gb2", "gwsa", "gbeta"):
        setattr(Bw, f, _fake())
    Bw.p, Bw.B, Bw.C, Bw.H, Bw.W, Bw.dtype = P, B, Cc, H, W, _lib.F32
    Bw.ctx_bytes, Bw.scratch_bytes = need_ctx, need_scr - 16
    assert lib.mgacbam_backward(bl, 1, None) == _lib.E_SIZE and b"scratch" in lib.mgacbam_last_error()
    Bw.ctx_bytes, Bw.scratch_bytes = 0, need_scr
    assert lib.mgacbam_backward(bl, 1, None) == _lib.E_SIZE and b"ctx" in lib.mgacbam_last_error()
    # MaskECA
    el = (_lib.EcaFwdLevel * 1)()
    E = el[0]
    E.x = E.mask = E.y = E.ctx = _fake()
    E.p = _lib.EcaParams(_fake(), _fake(), 3, 1, 1e-4, 1e-6)
    E.B, E.C, E.H, E.W, E.dtype = B, Cc, H, W, _lib.F32
    E.ctx_bytes = lib.mgacbam_eca_ctx_bytes(B, Cc, H, W) - 1
    assert lib.mgacbam_eca_forward(el, 1, None) == _lib.E_SIZE
    eb = (_lib.EcaBwdLevel * 1)()
    Eb = eb[0]
    for f in ("x", "mask", "gy", "ctx", "scratch", "gx", "gmask", "gw", "gbeta"):
        setattr(Eb, f, _fake())
    Eb.p, Eb.B, Eb.C, Eb.H, Eb.W, Eb.dtype = E.p, B, Cc, H, W, _lib.F32
    Eb.ctx_bytes, Eb.scratch_bytes = lib.mgacbam_eca_ctx_bytes(B, Cc, H, W), lib.mgacbam_eca_scratch_bytes(B, Cc, H, W) - 1
    assert lib.mgacbam_eca_backward(eb, 1, None) == _lib.E_SIZE and b"scratch" in lib.mgacbam_last_error()
    # mask head
    HP = _lib.HeadParams(*([_fake()] * 8), 16, 1e-3, 0.03, 1)
    hf = (_lib.HeadFwdLevel * 1)()
    Hf = hf[0]
    Hf.x = Hf.logits = Hf.ctx = _fake()
    Hf.p, Hf.B, Hf.C, Hf.H, Hf.W, Hf.dtype = HP, B, Cc, H, W, _lib.F32
    Hf.ctx_bytes = lib.mgahead_ctx_bytes(B, Cc, H, W, 16) - 4
    assert lib.mgahead_forward(hf, 1, None) == _lib.E_SIZE and b"ctx" in lib.mgacbam_last_error()
    hb = (_lib.HeadBwdLevel * 1)()
    Hb = hb[0]
    for f in ("x", "g_logits", "ctx", "scratch", "gx", "gw1", "gbn_weight", "gbn_bias", "gwh", "gbh"):
        setattr(Hb, f, _fake())
    Hb.p, Hb.B, Hb.C, Hb.H, Hb.W, Hb.dtype = HP, B, Cc, H, W, _lib.F32
    Hb.ctx_bytes, Hb.scratch_bytes = lib.mgahead_ctx_bytes(B, Cc, H, W, 16), lib.mgahead_bwd_scratch_bytes(B, Cc, H, W, 16) - 4
    assert lib.mgahead_backward(hb, 1, None) == _lib.E_SIZE and b"scratch" in lib.mgacbam_last_error()
    # segmentation loss
    sl = (_lib.SegLevel * 1)()
    S = sl[0]
    S.logits = S.target = S.glogits = _fake()
    S.B, S.H, S.W, S.Ht, S.Wt, S.dtype, S.scale_weight, S.resize = B, H, W, H, W, _lib.F32, 1.0, _lib.SEG_NEAREST
    cfg = _lib.SegCfg(1.0, 1.0, 1.0, 1.0, 0, 0.5, 0.6, 0.5)
    ws = lib.mgaseg_ws_bytes(sl, 1)
    assert ws > 0
    assert lib.mgaseg_forward(sl, 1, C.byref(cfg), _fake(), ws - 1, _fake(), None) == _lib.E_SIZE
    assert lib.mgaseg_backward(sl, 1, C.byref(cfg), _fake(), ws - 1, _fake(), None) == _lib.E_SIZE
    assert lib.mgaseg_kendall_forward(sl, 1, C.byref(cfg), _fake(), 0, _fake(), _fake(), 3, _fake(), _fake(), None) == _lib.E_SIZE
    assert b"ws" in lib.mgacbam_last_error()


def test_a_size_cached_across_a_knob_change_is_refused(built_lib, monkeypatch):
    """Round 2's abort (gpurun_out/r2b: rc 134): the binding cached mgacbam_bwd_scratch_bytes() across mgacbam_reload_env() although
    the launch geometry (tile counts) had changed -> undersized scratch -> out-of-bounds device writes.  The requirement is now
    recomputed under the current knobs inside every call and the stale capacity is refused."""
    from mga_yolo_amd import _lib
    lib = _lib.load()
    B, Cc, H, W, hid, k = 4, 64, 40, 40, 4, 7
    stale = lib.mgacbam_bwd_scratch_bytes(B, Cc, H, W, hid, k)            # under the default geometry
    monkeypatch.setenv("MGACBAM_CHAN_TX", "1")                            # one pixel vector per tile: many more tile partials
    lib.mgacbam_reload_env()                                              # (the raw entry point: _lib.reload_env() would also drop its size cache)
    try:
        fresh = lib.mgacbam_bwd_scratch_bytes(B, Cc, H, W, hid, k)
        assert fresh > stale
        bl = (_lib.BwdLevel * 1)()
        Bw = bl[0]
        for f in ("x", "mask", "gy", "ctx", "scratch", "gx", "gmask", "gw1", "gb1", "gw2", "gb2", "gwsa", "gbeta"):
            setattr(Bw, f, _fake())
        Bw.p = _lib.Params(*([_fake()] * 6), hid, k, 1, 1e-4, 1e-6)
        Bw.B, Bw.C, Bw.H, Bw.W, Bw.dtype = B, Cc, H, W, _lib.F32
        Bw.ctx_bytes, Bw.scratch_bytes = _lib.ctx_bytes(B, Cc, H, W, hid), stale
        assert lib.mgacbam_backward(bl, 1, None) == _lib.E_SIZE
        msg = lib.mgacbam_last_error().decode()
        assert "scratch" in msg and str(stale) in msg and str(fresh) in msg
    finally:
        monkeypatch.undo()
        _lib.reload_env()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No fallback: a device call with the library absent raises, it does not compute some other way."""
    import torch
    from mga_yolo_amd import _lib, functional
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope" / "libmgacbam.so"))
    assert not _lib.available()
    with pytest.raises(_lib.LibraryMissing, match="no fallback"):
        _lib.load()
    x = torch.zeros(1, 16, 4, 4)
    ps = [torch.zeros(1, 16), torch.zeros(1), torch.zeros(16, 1), torch.zeros(16), torch.zeros(1, 3, 7, 7), torch.zeros(())]
    with pytest.raises(_lib.LibraryMissing):
        functional.mask_cbam(x, None, *ps, functional.BlockConfig(hidden=1))


def test_host_tensor_never_reaches_the_device_entry_point(built_lib):
    import torch
    from mga_yolo_amd import functional
    x = torch.zeros(1, 16, 4, 4)
    ps = [torch.zeros(1, 16), torch.zeros(1), torch.zeros(16, 1), torch.zeros(16), torch.zeros(1, 3, 7, 7), torch.zeros(())]
    with pytest.raises(RuntimeError, match="device tensors only"):
        functional.mask_cbam(x, None, *ps, functional.BlockConfig(hidden=1))
