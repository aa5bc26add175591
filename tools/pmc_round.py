#!/usr/bin/env python3
"""Counters of the round's PMC passes -> profiles/rNN_pmc_raw.txt (every counter of every kernel, mean per dispatch) and
profiles/traffic.json (what bench.py attaches to its roofline blocks).

    python3 tools/pmc_round.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ pmc_SQ/ pmc_SQ2/> <out dir> <round number>

Kernels are told apart by name AND grid size (the stream copy that replicates inputs is the same kernel as the measured copy).
traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE counts 64 bytes per 128-byte request (MI355X_MICROARCH.md, HBM),
WRITE_SIZE is exact; FETCH_SIZE and WRITE_SIZE come from separate passes (they do not fit one)."""
import collections, csv, glob, json, os, sys

src, out = sys.argv[1], sys.argv[2]
ROUND = int(sys.argv[3]) if len(sys.argv) > 3 else 6
TAG = "r%02d" % ROUND
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))

# bench key -> (kernel name substring, grid size in threads or None, algorithmic bytes)
W = H = 8192
FR = 7680 * 4320 + 2 * 3840 * 2160
WANT = {
    "k_i16_roundtrip": ("k_i16_tile<2, false, true, 2>", None, 4 * W * H),
    "k_i16_roundtrip_lut": ("k_i16_tile<2, true, false, 2>", None, 4 * W * H),
    "k_i16_fwd": ("k_i16_tile<0, false, true, 2>", None, 4 * W * H),
    "k_i16_inv": ("k_i16_tile<1, false, true, 2>", None, 4 * W * H),
    "k_stream_copy": ("k_stream_copy", (W * H * 2 // 16 // 8), 4 * W * H),
    "k_q32_avx": ("k_q32_tile", None, 2 * W * H),
    "k_stereo_sse": ("k_fwd_quant_u8<1, 1, false, true>", None, 2 * W * H),
    "k_stereo_scalar": ("k_fwd_quant_u8<2, 1, false, true>", None, 2 * W * H),
    "k_encq_sse": ("k_fwd_quant_u8<1, 3, false, true>", None, 2 * W * H),
    "k_encq_scalar": ("k_fwd_quant_u8<2, 2, false, true>", None, 2 * W * H),
    "k_f32_tile_fwd": ("k_f32_tile<0>", None, 8 * W * H),
    "k_i16_batch_420": ("k_i16_batch<2, 1, false, false>", None, 4 * FR),
    "k_u8_batch_420": ("k_u8_batch<0, false, false>", None, 2 * FR),
    "k_q32_batch_420": ("k_q32_batch<false>", None, 2 * FR),
    "k_u8_batch_420_fwd": ("k_u8_batch<1, false, false>", None, 3 * FR),
    "k_u8_batch_420_inv": ("k_u8_batch<2, true, false>", 12150 * 64, 3 * FR),  # (12,150 tiles: the chroma planes' rows are tiled in pairs)
    "k_i16_batch_fwd_256": ("k_i16_batch<0, 1, true, false>", 256 * 4096 * 64, 4 * 256 * 4096 * 4096),
    "k_u8_i16_fwd": ("k_u8_i16_fwd", None, 3 * W * H),
    "k_u8_i16_inv": ("k_u8_batch<2, true, false>", (W // 8) * (H // 8), 3 * W * H),  # mdct_inv_i16_u8 = a batch of one
    "k_scan_i16_rle": ("k_scan<0, true>", None, 5 * W * H),
    "k_scan_q32_rle": ("k_scan<1, true>", None, 4 * W * H),
    "k_u8_records": ("k_u8_records<false, false>", None, 4 * W * H),
    "k_split420": ("k_split420<false>", None, 6 * W * H),
    "k_split420_u8_planes": ("k_split420<true>", None, 9 * W * H // 2),
    "k_huffman_rows_q60": ("k_huffman_rows", None, 3 * W * H),
    "k_px_huffman_rows_q60": ("k_px_huffman_rows<false, 4, false, false>", None, W * H),
    "k_px_jpeg_scan_k1": ("k_px_huffman_rows<false, 4, true, false>", None, W * H),
}


def find(sub, grid):
    hits = [k for k in acc if sub in k[0] and (grid is None or k[1] == grid)]
    if not hits:
        return None
    return max(hits, key=lambda k: sum(len(v) for v in acc[k].values()))  # (several grids: the one launched most)


lines, traffic = [], {"traffic_round": ROUND,
                      "source": f"profiles/{TAG}_pmc_raw.txt (round {ROUND}, tools/profile_round.sh on the round's final build): separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; two SQ sets) "
                                "of tools/run_kernel.py all -- every input uploaded or replicated by the library's own copy kernel, no torch kernels, the 256-plane batch at its real size; "
                                "bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB per dispatch (gfx950: FETCH_SIZE tallies 128-byte requests at 64)",
                      "kernels": {}}
for key in sorted(acc):
    lines.append(f"{key[0][:120]}   grid {key[1]}")
    for c in sorted(acc[key]):
        v = acc[key][c]
        lines.append(f"    {c:26s} n={len(v):3d} mean={sum(v) / len(v):.6g}")
for name, (sub, grid, alg) in WANT.items():
    k = find(sub, grid)
    if k is None:
        continue
    m = {c: sum(v) / len(v) for c, v in acc[k].items()}
    e = {"kernel": k[0], "grid_threads": k[1], "algorithmic_bytes_per_launch": alg}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["bytes_per_launch"] = int(round((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024))
        e["traffic_over_algorithmic"] = round(e["bytes_per_launch"] / alg, 4)
    if "SQ_WAVES" in m and m["SQ_WAVES"]:
        e["waves"] = int(round(m["SQ_WAVES"]))
        for c, short in (("SQ_INSTS_VALU", "valu_insts_per_wave"), ("SQ_INSTS_SALU", "salu_insts_per_wave"), ("SQ_INSTS_LDS", "lds_insts_per_wave"),
                         ("SQ_INSTS_VMEM_RD", "vmem_rd_per_wave"), ("SQ_INSTS_VMEM_WR", "vmem_wr_per_wave")):
            if c in m:
                e[short] = round(m[c] / m["SQ_WAVES"], 1)
        for c in ("SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            if c in m:
                e[c] = round(m[c], 1)
    traffic["kernels"][name] = e
    traffic[name + "_bytes_per_launch"] = e.get("bytes_per_launch")  # (flat keys: what earlier rounds' bench.py read)
os.makedirs(out, exist_ok=True)
open(os.path.join(out, TAG + "_pmc_raw.txt"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
for name, e in traffic["kernels"].items():
    print(f"{name:24s} bytes {e.get('bytes_per_launch')}  x{e.get('traffic_over_algorithmic')}  valu/wave {e.get('valu_insts_per_wave')}  waves {e.get('waves')}")
