"""Deterministic synthetic planes: the same counter hash on host (numpy) and
device (torch), so CPU oracle and GPU path see bit-identical inputs without any files.

    mix32(x): x *= 0x9E3779B1; x ^= x >> 15; x *= 0x85EBCA77; x ^= x >> 13     (mod 2^32)
    h(i, seed) = mix32(i ^ seed)
    noise : px = h >> 24
    photo : b = ((x*3 + y*5) >> 2) & 0xFF; tri = b < 128 ? b : 255 - b
            px = clamp(48 + tri + ((h >> 24) % 49) - 24, 0, 255)
    int16 : px - 128 (8-bit) or (h >> 20) - 2048 (12-bit)
"""
import numpy as np

SEED = 20261003
_M = 0xFFFFFFFF


def _mix32_np(x):
    x = (x * np.uint64(0x9E3779B1)) & np.uint64(_M)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x85EBCA77)) & np.uint64(_M)
    x ^= x >> np.uint64(13)
    return x


def hash_np(n, seed=SEED, offset=0):
    i = np.arange(offset, offset + n, dtype=np.uint64)
    return _mix32_np((i ^ np.uint64(seed & _M)) & np.uint64(_M))


def plane_u8_np(W, H, kind="noise", seed=SEED):
    h = hash_np(W * H, seed)
    if kind == "noise":
        return (h >> np.uint64(24)).astype(np.uint8).reshape(H, W)
    i = np.arange(W * H, dtype=np.uint64)
    x, y = i % np.uint64(W), i // np.uint64(W)
    b = ((x * np.uint64(3) + y * np.uint64(5)) >> np.uint64(2)) & np.uint64(0xFF)
    tri = np.where(b < 128, b, np.uint64(255) - b).astype(np.int64)
    px = 48 + tri + ((h >> np.uint64(24)) % np.uint64(49)).astype(np.int64) - 24
    return np.clip(px, 0, 255).astype(np.uint8).reshape(H, W)


def plane_i16_np(W, H, kind="photo", seed=SEED, bits=8):
    if bits == 12:
        return ((hash_np(W * H, seed) >> np.uint64(20)).astype(np.int32) - 2048).astype(np.int16).reshape(H, W)
    return (plane_u8_np(W, H, kind, seed).astype(np.int16) - 128)


def _mix32_t(x):
    x = (x * 0x9E3779B1) & _M
    x = x ^ (x >> 15)
    x = (x * 0x85EBCA77) & _M
    x = x ^ (x >> 13)
    return x


def plane_u8_torch(W, H, kind="noise", seed=SEED, device="cuda"):
    import torch

    i = torch.arange(W * H, dtype=torch.int64, device=device)
    h = _mix32_t(((i ^ (seed & _M)) & _M))
    if kind == "noise":
        return (h >> 24).to(torch.uint8).reshape(H, W)
    x, y = i % W, i // W
    b = ((x * 3 + y * 5) >> 2) & 0xFF
    tri = torch.where(b < 128, b, 255 - b)
    px = 48 + tri + ((h >> 24) % 49) - 24
    return px.clamp_(0, 255).to(torch.uint8).reshape(H, W)


def plane_i16_torch(W, H, kind="photo", seed=SEED, bits=8, device="cuda"):
    import torch

    if bits == 12:
        i = torch.arange(W * H, dtype=torch.int64, device=device)
        h = _mix32_t(((i ^ (seed & _M)) & _M))
        return ((h >> 20) - 2048).to(torch.int16).reshape(H, W)
    return plane_u8_torch(W, H, kind, seed, device).to(torch.int16) - 128


# ITU-T T.81 Annex K.1 quantisation tables (Tables K.1 luminance, K.2 chrominance), natural order v*8+u: the per-plane
# tables of BASELINE.json configs[2] (bench.py, tests, tools/ share these literals)
JPEG_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                      18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
JPEG_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, dtype=np.float32)

# BASELINE.json configs[2]: the 8K 4:2:0 frame -- (width, height, seed offset, table) of Y, Cb, Cr
CONFIG3_PLANES = ((7680, 4320, 0, "luma"), (3840, 2160, 1, "chroma"), (3840, 2160, 2, "chroma"))
