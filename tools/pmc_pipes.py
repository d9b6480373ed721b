"""Per-kernel pipe utilisation from one SQ counter pass (tools/profile_r05.sh -> pmc/sq.csv):

    matrix pipe busy   = SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles
    vector issue busy  = 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles           (quad-cycle counter)
    waves waiting      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES               (share of wave lifetime spent waiting on an instruction)
    SIMD-cycles        = GRBM_GUI_ACTIVE / 8 (XCDs) x 1024 SIMDs         (MI355X_MICROARCH.md: rocprofv3 sums the XCDs)

All counters are sums over the dispatches of a kernel in the pass."""
import csv
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
out = {}
for r in rows:
    g = float(r.get("GRBM_GUI_ACTIVE", 0) or 0)
    if g <= 0:
        continue
    simd_cycles = g / 8.0 * 1024.0
    f = lambda k: float(r.get(k, 0) or 0)   # noqa: E731
    wave = f("SQ_WAVE_CYCLES")
    out[r["kernel"]] = {
        "dispatches": int(float(r["dispatches"])),
        "matrix_pipe_busy": round(f("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles, 4),
        "vector_issue_busy": round(4.0 * f("SQ_ACTIVE_INST_VALU") / simd_cycles, 4),
        "any_issue_busy": round(4.0 * f("SQ_ACTIVE_INST_ANY") / simd_cycles, 4),
        "waves_waiting_on_an_instruction": round(f("SQ_WAIT_INST_ANY") / wave, 4) if wave else None,
        "valu_instructions": f("SQ_INSTS_VALU"),
        "mean_waves_resident_per_simd": round(4.0 * wave / simd_cycles, 3) if wave else None,
    }
json.dump(out, open(dst, "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["matrix_pipe_busy"])[:12]:
    print(f"{k[:60]:60s} matrix {v['matrix_pipe_busy']:.3f} vector {v['vector_issue_busy']:.3f} waiting {v['waves_waiting_on_an_instruction']}")
