"""CPU: libvloam_hip.so loads without a GPU, exports every symbol include/vloam_hip/c_api.h declares, validates its
arguments, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "vloam_hip", "c_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vloam_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(vl):
    L = vl.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libvloam_hip.so does not export %s" % n


def test_default_config_matches_reference_launch_files(vl):
    c = vl.default_config()
    # LOM/launch/loam_velodyne_HDL_64_kitti.launch:3-16, MAIN/launch/vloam_main.launch:4-6
    assert c.scan_line == 64 and c.minimum_range == 5.0 and c.mapping_skip_frame == 1
    assert abs(c.mapping_line_resolution - 0.4) < 1e-7 and abs(c.mapping_plane_resolution - 0.8) < 1e-7
    assert c.detach_VO_LO == 1 and c.reset_VO_to_identity == 0 and c.remove_VO_outlier == 100


def test_create_validates_and_has_no_cpu_fallback(vl):
    L = vl.lib()
    h = C.c_void_p()
    bad = vl.default_config(scan_line=48)
    assert L.vloam_create(C.byref(bad), 0, C.byref(h)) == vl.ERR_INVALID     # reference: ROS_BREAK() on a bad scan_line
    assert b"16, 32 or 64" in L.vloam_last_error()
    import torch
    st = L.vloam_create(C.byref(vl.default_config()), 0, C.byref(h))
    if torch.cuda.is_available():
        assert st == vl.VLOAM_OK
        L.vloam_destroy(h)
    else:
        assert st == vl.ERR_NO_DEVICE and b"no CPU fallback" in L.vloam_last_error()
    assert L.vloam_destroy(None) == vl.VLOAM_OK
    assert L.vloam_reset_frame(None) == vl.ERR_INVALID
    # argument checks that need no device: batch size, mapping leaf (75 x radix^3 voxel positions in a 32-bit tie rank: >= 0.132 m)
    assert L.vloam_create_batch(C.byref(vl.default_config()), 0, 0, C.byref(h)) == vl.ERR_INVALID
    assert L.vloam_create_batch(C.byref(vl.default_config()), 0, 1000, C.byref(h)) == vl.ERR_INVALID and b"n_sessions" in L.vloam_last_error()
    assert L.vloam_create(C.byref(vl.default_config(mapping_line_resolution=0.13)), 0, C.byref(h)) == vl.ERR_INVALID and b"0.132" in L.vloam_last_error()
    assert L.vloam_create(C.byref(vl.default_config(mapping_plane_resolution=0.0)), 0, C.byref(h)) == vl.ERR_INVALID
    assert L.vloam_select_session(None, 0) == vl.ERR_INVALID and L.vloam_batch_size(None, None) == vl.ERR_INVALID
    L.vloam_profile_kernel_name.restype = C.c_char_p
    names = [L.vloam_profile_kernel_name(k).decode() for k in range(L.vloam_profile_kernel_count())]
    assert "k_lm_solve" in names and "k_map_assoc" in names and names[0] == ""


def test_struct_layouts_match_the_device_header(vl):
    """ctypes mirrors of vloam_config / LMRecord stay in sync with the C structs."""
    text = open(os.path.join(ROOT, "include", "vloam_hip", "c_api.h")).read()
    body = re.search(r"typedef struct vloam_config \{(.*?)\} vloam_config;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int|double|float)\s+([a-zA-Z_0-9]+);", body)
    assert fields == [f[0] for f in vl.Config._fields_]
    assert C.sizeof(vl.LMRecord) == 8 * (7 + 7 + 36 + 6 + 6 + 4 + 8 * vl.K_LM_MAX_TRACE)
