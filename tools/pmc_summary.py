"""profiles/<round>/pmc/{FETCH_SIZE,WRITE_SIZE}.csv (tools/pmc_passes.sh, separate rocprofv3 --pmc passes) ->
profiles/pmc_summary.json: HBM-side bytes per SAMPLE for each C-ABI entry point, which bench.py attaches to its
roofline object as ``traffic`` (bytes per launch = bytes per sample x samples per launch).

Corrections, per MI355X_MICROARCH.md "HBM": FETCH_SIZE is reported in KiB-like units of 1024 B by rocprofv3 here
(counter x 1024 = bytes is what the kernel_stats of round 1 cross-checked) and counts 64 B per 128-B request on
gfx950 -> doubled; WRITE_SIZE reads exact.

    python tools/pmc_summary.py profiles/r02b_.../pmc <samples in the profiled run> profiles/pmc_summary.json [build tag]
"""
import csv
import json
import os
import sys

ENTRY = {  # kernel name fragment -> C-ABI entry point (first match wins; <true> = taps derived in-kernel, the _pts entry points)
    "fd7_fwd_kernel<true, true>": "rsdf_hashgrid_fwd_fd7_x2",
    "fd7_fwd_kernel<false, true>": "rsdf_hashgrid_fwd_fd7_x2",
    "fd7_fwd_kernel<true, false>": "rsdf_hashgrid_fwd_fd7_pts",
    "fwd_x2_kernel": "rsdf_sdfmlp_fd7_fwd_x2",
    "bwd_x2_kernel": "rsdf_sdfmlp_fd7_bwd_x2",
    "absmax_kernel": "rsdf_sdfmlp_fd7_bwd_x2",
    "fd7_fwd_kernel<true>": "rsdf_hashgrid_fwd_fd7_pts",
    "fd7_produce_kernel<true>": "rsdf_hashgrid_bwd_fd7_pts",
    "fd7_produce_kernel<true,": "rsdf_hashgrid_bwd_fd7_pts",     # round 6: <DERIVE, REC> (the record format is a template argument)
    "fd7_reduce_kernel": "rsdf_hashgrid_bwd_fd7_pts",       # (bench.py only runs the _pts form)
    "fd7_fwd_kernel": "rsdf_hashgrid_fwd_fd7",
    "fd7_produce_kernel": "rsdf_hashgrid_bwd_fd7",
    "coop_bwd_kernel": "rsdf_sdfmlp_fd7_bwd",
    "quad_bwd_kernel": "rsdf_sdfmlp_fd7_bwd",
    "sdfmlp_bwd_kernel": "rsdf_sdfmlp_fd7_bwd",
    "coop_fwd_kernel": "rsdf_sdfmlp_fd7_fwd",
    "sdfmlp_fwd_kernel": "rsdf_sdfmlp_fd7_fwd",
    "hashgrid_fwd_kernel": "rsdf_hashgrid_fwd",
    "hashgrid_bwd_kernel": "rsdf_hashgrid_bwd",
}


def read(path, counter):
    out = {}
    if not os.path.exists(path):
        return out
    for row in csv.DictReader(open(path)):
        for frag, entry in ENTRY.items():
            if frag in row["kernel"]:
                out[entry] = out.get(entry, 0.0) + float(row[counter])
                break
    return out


def main():
    src, samples, dst = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    build = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    fetch = read(os.path.join(src, "FETCH_SIZE.csv"), "FETCH_SIZE")
    write = read(os.path.join(src, "WRITE_SIZE.csv"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, 0.0) * 1024.0 * 2.0        # KiB units; gfx950 tallies 128-B requests at 64 B
        w = write.get(k, 0.0) * 1024.0
        kernels[k] = {"fetch_bytes_per_sample": f / samples, "write_bytes_per_sample": w / samples,
                      "hbm_bytes_per_sample": (f + w) / samples}
    json.dump({"source": f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md)", "build": build, "samples_in_profiled_run": samples, "kernels": kernels},
              open(dst, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:28s} fetch {v['fetch_bytes_per_sample']:9.0f}  write {v['write_bytes_per_sample']:9.0f}  B/sample")


if __name__ == "__main__":
    main()
