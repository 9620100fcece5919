#!/usr/bin/env python3
"""Turn a tools/pmc_summary.py text summary into profiles/<round>_traffic.json for one kernel: HBM bytes per launch
from FETCH_SIZE / WRITE_SIZE (collected in separate passes), with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section: FETCH_SIZE is in KiB and under-counts 128-B requests as 64 B -> x2
upper bound; WRITE_SIZE in KiB as is).
usage: make_traffic_json.py <pmc_summary.txt> <kernel-substring> <algorithmic-bytes> <note> <commit> <out.json>"""
import json
import sys


def main():
    path, kern, alg, note, commit, out = sys.argv[1:7]
    vals, name, cur = {}, None, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            if kern in cur and name is None:
                name = cur
            continue
        if cur == name and name is not None:
            f = line.split()
            vals[f[0]] = float(f[1])
            vals["_n_" + f[0]] = int(f[2].strip("(x)"))
    if name is None or "FETCH_SIZE" not in vals:
        sys.exit("kernel %r or FETCH_SIZE not found in %s" % (kern, path))
    fetch_kb, write_kb = vals["FETCH_SIZE"], vals.get("WRITE_SIZE", 0.0)
    doc = {
        "kernel": kern,
        "kernel_symbol": name,
        "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024.0, -3)),
        "fetch_size_kb": fetch_kb,
        "fetch_correction": "x2 (gfx950: FETCH_SIZE counts 128-B requests at 64 B; MI355X_MICROARCH.md, HBM section)",
        "write_size_kb": write_kb,
        "launches_averaged": vals["_n_FETCH_SIZE"],
        "algorithmic_bytes_per_launch": int(alg),
        "algorithmic_bytes_note": note,
        "source": "%s (rocprofv3 --pmc, one counter group per pass, tools/pmc_workload.py 1000 3)" % path,
        "commit": commit,
    }
    for k_in, k_out in (("TCC_HIT_sum", "tcc_hit"), ("TCC_MISS_sum", "tcc_miss"), ("SQ_INSTS_VALU_MFMA_MOPS_I8", "mfma_mops_i8_x512"),
                        ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_cycles"), ("GRBM_GUI_ACTIVE", "grbm_gui_active_sum_over_8_xcds"),
                        ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles"), ("SQ_LDS_IDX_ACTIVE", "lds_idx_active_cycles")):
        if k_in in vals:
            doc[k_out] = vals[k_in]
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
