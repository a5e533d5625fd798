"""HBM-side traffic per tg_ns_homo_batched(_ws) launch from rocprofv3 --pmc passes (separate read / write passes, as
MI355X_MICROARCH.md prescribes): bytes = 128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B + 64*WRREQ_64B + 32*(WRREQ -
WRREQ_64B), summed over every kernel of the launch (window-ordered form: win_init + per hop win_emit, win_hist,
win_colscan, win_basescan, win_scatter, win_gather; fused form: ns_homo_uniform_kernel) and divided by the launches.
usage: pmc_traffic.py <rd_dir> <wr_dir> <form> <batches_per_launch> -> JSON"""
import csv
import glob
import json
import sys
from collections import defaultdict

rd_dir, wr_dir, form, bpl = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
pipeline = sys.argv[5] if len(sys.argv) > 5 else "push"
needles = ("tg::win_",) if form == "windowed" else ("ns_homo_uniform_kernel",)
# the first kernel of a launch: win_init (round 2's pipeline), win_first_hops* (fused first hops), win_vtab (staged form)
launch_markers = ("win_init_kernel", "win_first_hops", "win_vtab_kernel") if form == "windowed" else ("ns_homo_uniform_kernel",)


def collect(root):
    tot, per_kernel, launches = defaultdict(float), defaultdict(lambda: defaultdict(float)), set()
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            if any(nd in name for nd in needles):
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                short = name.split("(")[0].split("<")[0].replace("void ", "")
                per_kernel[short][row["Counter_Name"]] += float(row["Counter_Value"])
                if any(mk in name for mk in launch_markers):
                    launches.add(row["Dispatch_Id"])
    return tot, per_kernel, max(len(launches), 1)


rd, rd_k, n_rd = collect(rd_dir)
wr, wr_k, n_wr = collect(wr_dir)
rd_bytes = (128 * rd["TCC_EA0_RDREQ_128B_sum"] + 64 * rd["TCC_EA0_RDREQ_64B_sum"] + 32 * rd["TCC_EA0_RDREQ_32B_sum"]) / n_rd
wr_bytes = (64 * wr["TCC_EA0_WRREQ_64B_sum"] + 32 * (wr["TCC_EA0_WRREQ_sum"] - wr["TCC_EA0_WRREQ_64B_sum"])) / n_wr
out = {
    "form": form, "pipeline": pipeline, "batches_per_launch": bpl, "idx32": 1, "ptr32": 1, "launches_profiled": [n_rd, n_wr],
    "read_requests_per_launch": rd["TCC_EA0_RDREQ_sum"] / n_rd,
    "read_requests_128B_per_launch": rd["TCC_EA0_RDREQ_128B_sum"] / n_rd,
    "write_requests_per_launch": wr["TCC_EA0_WRREQ_sum"] / n_wr,
    "write_requests_64B_per_launch": wr["TCC_EA0_WRREQ_64B_sum"] / n_wr,
    "read_bytes_per_launch": rd_bytes, "write_bytes_per_launch": wr_bytes,
    "hbm_bytes_per_launch": rd_bytes + wr_bytes,
    "per_kernel_read_requests_per_launch": {k: v["TCC_EA0_RDREQ_sum"] / n_rd for k, v in sorted(rd_k.items())},
    "per_kernel_write_requests_per_launch": {k: v["TCC_EA0_WRREQ_sum"] / n_wr for k, v in sorted(wr_k.items())},
}
print(json.dumps(out, indent=1))
