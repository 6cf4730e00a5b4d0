#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of bench.py (FETCH_SIZE pass, WRITE_SIZE pass) into
profiles/<tag>_hbm_traffic.{md,json} and profiles/hbm_traffic_latest.json.

usage: tools/summarize_pmc.py <fetch_dir> <write_dir> <tag> <nx> <ny> "<config note>" [<sq_dir>]

With <sq_dir> (a pass with SQ_INSTS_VALU ...) the VALU wave-instructions per launch are recorded too.  The JSON carries
a hash of the kernel sources so that bench.py can tell when a committed profile no longer belongs to the code.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE (KB) reads exactly 1/2 of the bytes of a
coalesced stream -- re-checked here on three kernels of the same run whose traffic is known exactly
(8 B/lane unit-stride loads, the access pattern of all kernels in this repo); WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    f = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)  # the newest pass in that directory
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = None
        for k in ("mevp_fused_kernel", "mevp_fused2_kernel", "mevp_fused3_kernel", "mevp_fused4_kernel", "transport_stage_kernel<2>", "transport_pair_kernel<2>", "transport_march_kernel<2>", "mevp_prepare_kernel", "wind_stress_kernel", "ice_strength_kernel",
                  "mevp_pack_nodal_kernel", "mevp_stress_kernel", "mevp_velocity_kernel", "column_step_kernel"):
            if k in name:
                key = k
        if key:
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def source_hash():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench

    return bench.kernel_source_hash()


def main():
    fdir, wdir, tag, nx, ny, note = sys.argv[1:7]
    nx, ny = int(nx), int(ny)
    fe, wr = load(fdir), load(wdir)
    sq = load(sys.argv[7]) if len(sys.argv) > 7 else {}
    N, nn = nx * ny, (2 * nx + 1) * (2 * ny + 1)
    mean = lambda a: sum(a) / len(a)
    known = {"wind_stress_kernel": (2 * 8 * nn, 2 * 8 * nn), "ice_strength_kernel": (12 * 8 * N, 9 * 8 * N),
             "mevp_pack_nodal_kernel": (8 * 8 * nn, 6 * 8 * nn)}
    L = ["# HBM traffic, %dx%d, %s\n\n" % (nx, ny, note),
         "Two separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`) of `rocprofv3 --kernel-trace --output-format csv -- python bench.py "
         "--steps 1 --warmup 0 --nsub 8 --no-cpu-baseline`; counter unit KB.\n\n"
         "Calibration in this access pattern (8 B/lane unit-stride) on kernels with exactly known traffic:\n\n"
         "| kernel | known read KB | FETCH_SIZE | ratio | known write KB | WRITE_SIZE | ratio |\n|---|---|---|---|---|---|---|\n"]
    for k, (r, w) in known.items():
        if k in fe and k in wr:
            f, ws = mean(fe[k]["FETCH_SIZE"]), mean(wr[k]["WRITE_SIZE"])
            L.append("| %s | %.4g | %.4g | %.3f | %.4g | %.4g | %.3f |\n" % (k, r / 1024, f, f / (r / 1024), w / 1024, ws, ws / (w / 1024)))
    L.append("\n=> read bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction), write bytes = WRITE_SIZE x 1024.\n\n"
             "| kernel | launches | read GB | write GB | total GB / launch | algorithmic GB / launch |\n|---|---|---|---|---|---|\n")
    out = {}
    for k, alg in (("mevp_fused_kernel", 896 * N), ("mevp_fused2_kernel", 2 * 896 * N), ("mevp_fused3_kernel", 3 * 896 * N), ("mevp_fused4_kernel", 4 * 896 * N), ("mevp_stress_kernel", None), ("mevp_velocity_kernel", None),
                   ("transport_stage_kernel<2>", 1008 * N / 3), ("transport_pair_kernel<2>", 1008 * N / 3), ("column_step_kernel", 160 * N)):
        if k in fe and k in wr:
            rd, w = 2 * mean(fe[k]["FETCH_SIZE"]) * 1024, mean(wr[k]["WRITE_SIZE"]) * 1024
            L.append("| %s | %d | %.3f | %.3f | %.3f | %s |\n" % (k, len(fe[k]["FETCH_SIZE"]), rd / 1e9, w / 1e9, (rd + w) / 1e9,
                                                                  "%.3f" % (alg / 1e9) if alg else "-"))
            out[k] = {"read_bytes": rd, "write_bytes": w, "total_bytes": rd + w, "algorithmic_bytes": alg}
            if k in sq and "SQ_INSTS_VALU" in sq[k]:
                out[k]["valu_wave_insts"] = mean(sq[k]["SQ_INSTS_VALU"])
                for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVES"):
                    if c in sq[k]:
                        out[k][c] = mean(sq[k][c])
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    open(os.path.join(root, tag + "_hbm_traffic.md"), "w").write("".join(L))
    js = {"nx": nx, "ny": ny, "config": note, "file": tag + "_hbm_traffic.json", "kernel_source_hash": source_hash(), "kernels": out}
    json.dump(js, open(os.path.join(root, tag + "_hbm_traffic.json"), "w"), indent=1)
    json.dump(js, open(os.path.join(root, "hbm_traffic_latest.json"), "w"), indent=1)
    print("".join(L))


if __name__ == "__main__":
    main()
