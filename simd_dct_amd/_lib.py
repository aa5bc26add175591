"""ctypes binding of libmdct_hip.so -- the C-ABI declared in include/mdct.h.

The HIP library is the product.  There is NO fallback: if the shared object is missing
or fails to load, every entry point raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MDCT_LIB_PATH: another build of the same library (A/B timing of kernel variants, tools/); never a different product
LIB_PATH = os.environ.get("MDCT_LIB_PATH") or os.path.join(_HERE, "libmdct_hip.so")

c_size_t = ctypes.c_size_t
c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
f32p = ctypes.POINTER(ctypes.c_float)


class DeviceInfo(ctypes.Structure):
    """mdct_device_info (include/mdct.h)."""

    _fields_ = [
        ("device", c_int),
        ("compute_units", c_int),
        ("wavefront_size", c_int),
        ("lds_bytes_per_cu", c_int),
        ("is_gfx950", c_int),
        ("hbm_bytes", c_size_t),
        ("name", ctypes.c_char * 128),
    ]


class PlaneI16(ctypes.Structure):
    """mdct_plane_i16 (include/mdct.h)."""

    _fields_ = [
        ("src", c_void_p),
        ("dst", c_void_p),
        ("pitch_in", c_size_t),
        ("pitch_out", c_size_t),
        ("sizeX", c_size_t),
        ("sizeY", c_size_t),
        ("lut", f32p),
    ]


PlaneU8 = PlaneI16  # mdct_plane_u8: the same layout, pitches in bytes

_PLANE = [c_void_p, c_void_p, c_size_t, c_size_t, f32p, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]
_PLANE_F32 = [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]
_REF = [c_int, c_void_p, c_void_p, f32p, c_size_t, c_size_t, c_size_t, c_size_t]

# name -> (restype, argtypes); every function include/mdct.h and include/simd_dct_shim.h declare
SIGNATURES = {
    "mdct_init": (c_int, [c_int]),
    "mdct_get_device_info": (c_int, [ctypes.POINTER(DeviceInfo)]),
    "mdct_last_error": (ctypes.c_char_p, []),
    "mdct_fwd_quant_u8": (c_int, [c_void_p, c_void_p, c_size_t, f32p, c_size_t, c_size_t, c_size_t, c_size_t, c_int, c_int, c_void_p]),
    "mdct_fwd_quant_u8_pitched": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, f32p, c_size_t, c_size_t, c_size_t, c_size_t, c_int, c_int, c_void_p]),
    "mdct_fwd_i16": (c_int, _PLANE),
    "mdct_inv_i16": (c_int, _PLANE),
    "mdct_roundtrip_i16": (c_int, _PLANE),
    "mdct_fwd_u8_records": (c_int, [c_void_p, c_size_t, f32p, c_int, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdct_fwd_i16_records": (c_int, [c_void_p, c_size_t, f32p, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdct_fwd_u8_i16": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, f32p, c_int, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mdct_inv_i16_u8": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, f32p, c_int, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mdct_fwd_f32": (c_int, _PLANE_F32),
    "mdct_inv_f32": (c_int, _PLANE_F32),
    "mdct_roundtrip_i16_planes": (c_int, [ctypes.POINTER(PlaneI16), c_int, c_void_p]),
    "mdct_fwd_i16_batch": (c_int, [ctypes.POINTER(PlaneI16), c_int, c_void_p]),
    "mdct_inv_i16_batch": (c_int, [ctypes.POINTER(PlaneI16), c_int, c_void_p]),
    "mdct_roundtrip_i16_batch": (c_int, [ctypes.POINTER(PlaneI16), c_int, c_void_p]),
    "mdct_batch_create": (c_int, [ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(PlaneI16), c_int]),
    "mdct_batch_run": (c_int, [c_void_p, c_void_p]),
    "mdct_roundtrip_u8": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, f32p, c_int, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mdct_roundtrip_u8_batch": (c_int, [ctypes.POINTER(PlaneU8), c_int, c_int, c_void_p]),
    "mdct_fwd_u8_i16_batch": (c_int, [ctypes.POINTER(PlaneU8), c_int, c_int, c_void_p]),
    "mdct_inv_i16_u8_batch": (c_int, [ctypes.POINTER(PlaneU8), c_int, c_int, c_void_p]),
    "mdct_batch_create_u8_i16": (c_int, [ctypes.POINTER(c_void_p), c_int, ctypes.POINTER(PlaneU8), c_int, c_int]),
    "mdct_batch_create_u8": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(PlaneU8), c_int, c_int]),
    "mdct_fwd_quant32_u8_batch": (c_int, [ctypes.POINTER(PlaneU8), c_int, c_void_p]),
    "mdct_batch_create_q32": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(PlaneU8), c_int]),
    "mdct_batch_launches": (c_int, [c_void_p]),
    "mdct_batch_destroy": (c_int, [c_void_p]),
    "mdct_zigzag_rle_i16": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdct_zigzag_rle_q32": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdct_zigzag_rle_u8": (c_int, [c_void_p, c_int, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdct_huffman_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mdct_fwd_u8_huffman_rows": (c_int, [c_void_p, c_size_t, f32p, c_int, c_size_t, c_size_t, c_size_t, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mdct_fwd_i16_huffman_rows": (c_int, [c_void_p, c_size_t, f32p, c_size_t, c_size_t, c_size_t, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "mdct_fwd_u8_jpeg_scan": (c_int, [c_void_p, c_size_t, f32p, c_int, c_size_t, c_size_t, c_size_t, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mdct_fwd_i16_jpeg_scan": (c_int, [c_void_p, c_size_t, f32p, c_size_t, c_size_t, c_size_t, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mdct_jpeg_pack_rows_counted": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mdct_jpeg_pack_rows": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mdct_huffman_seg_stride": (c_size_t, [c_size_t]),
    "mdct_huffman_spec": (c_int, [c_int, c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    "mdct_zigzag_table": (None, [c_void_p]),
    "mdct_split420_u8": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "mdct_split420_u8_planes": (c_int, [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "mdct_shard_rows": (None, [c_size_t, c_int, c_int, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    "mdct_stereo_shard_piece": (c_int, [c_size_t, c_size_t, c_int, c_int, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    "mdct_comm_get_unique_id": (c_int, [c_void_p]),
    "mdct_comm_init": (c_int, [ctypes.POINTER(c_void_p), c_int, c_int, c_void_p]),
    "mdct_comm_destroy": (c_int, [c_void_p]),
    "mdct_comm_rank": (c_int, [c_void_p]),
    "mdct_comm_world": (c_int, [c_void_p]),
    "mdct_allgather_rows": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "mdct_allgather_stereo": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "mdct_table_cache_stats": (c_int, [ctypes.POINTER(ctypes.c_uint64), c_int]),
    "mdct_clock_probe": (c_int, [c_void_p, ctypes.c_uint32, ctypes.c_uint32, c_void_p]),
    "mdct_stream_copy": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdct_timer_create": (c_void_p, []),
    "mdct_timer_destroy": (None, [c_void_p]),
    "mdct_timer_start": (c_int, [c_void_p, c_void_p]),
    "mdct_timer_stop": (c_int, [c_void_p, c_void_p]),
    "mdct_timer_elapsed_ms": (ctypes.c_double, [c_void_p]),
    "mdct_timer_wait_spin": (c_int, [c_void_p]),
    "mdct_stream_synchronize": (c_int, [c_void_p]),
    "mdct_shim_set_max_simd": (None, [c_int]),
    "mdct_shim_get_max_simd": (c_int, []),
    "mdct_shim_set_stream": (None, [c_void_p]),
    "mdct_shim_set_async": (None, [c_int]),
    "mdct_shim_release": (None, []),
    "mdct_shim_warmup": (c_int, [c_size_t]),
    "mdct_shim_pin": (c_int, [c_void_p, c_size_t]),
    "mdct_shim_unpin": (c_int, [c_void_p]),
    "mdct_shim_call": (c_int, _REF),
    "mdct_shim_call_on": (c_int, _REF + [c_void_p, c_int]),
}

# the reference's three C++-linkage entry points (simd_dct.h:29-31), Itanium-mangled
MANGLED = (
    "_Z37simdDCT_EncodeQuantize32ReorderBufferPKhPhPKfmmmm",
    "_Z41simdDCT_EncodeQuantizeReorderStereoBufferPKhPhPKfmmmm",
    "_Z28simdDCT_EncodeQuantizeBufferPKhPhPKfmmmm",
)

_lib = None


def load():
    """Load libmdct_hip.so (once).  torch is imported first when available so that the
    library binds to the SAME libamdhip64 the process's tensors live in."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  simd_dct_amd has no CPU fallback."
        )
    if os.environ.get("MDCT_NO_TORCH_PRELOAD") != "1":  # host-only users (e.g. the CPU multi-rank test's children) skip the import
        try:
            import torch  # noqa: F401  (side effect: loads torch's HIP runtime before ours resolves)
        except Exception:
            pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
