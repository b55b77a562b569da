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
                 proj=B * hid * H * W if hid <= _lib.PROJ_MAX_HIDDEN else 0, sync=3 * B * ((H * W + 15) // 16 + 1) + 4 + B)
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
    lv = (_lib.FwdLevel * 1)()
    assert lib.mgacbam_forward(lv, 1, None) == -1            # MGACBAM_E_NULL: x / y / ctx missing
    assert lib.mgacbam_forward(lv, 0, None) == -5            # MGACBAM_E_LEVELS
    assert lib.mgacbam_forward(None, 1, None) == -1
    bl = (_lib.BwdLevel * 1)()
    assert lib.mgacbam_backward(bl, 1, None) == -1
    assert lib.mgacbam_resize_nearest(None, None, 1, 4, 4, 2, 2, None) == -1
    with pytest.raises(RuntimeError, match="argument error"):
        _lib.check(-2, "x")


def test_struct_mirrors_have_the_header_field_order():
    from mga_yolo_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\}" % struct, src, re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = names[0].split()[-1].lstrip("*")
            out.append(first)
            out += [n.strip().lstrip("*") for n in names[1:]]
        return out

    assert fields("mgacbam_params") == [f[0] for f in _lib.Params._fields_]
    assert fields("mgacbam_fwd_level") == [f[0] for f in _lib.FwdLevel._fields_]
    assert fields("mgacbam_bwd_level") == [f[0] for f in _lib.BwdLevel._fields_]
    assert fields("mgacbam_ctx_layout") == [f[0] for f in _lib.CtxLayout._fields_]


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
