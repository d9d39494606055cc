"""Build profiles/pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs with
--kernel-trace only, as MI355X_MICROARCH.md prescribes).  usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Values are KB per dispatch (x1024).  gfx950 correction: FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled
(calibrated on rmsprop_kernel: 3 reads + 2 writes of the flat parameter vector)."""
import collections, csv, glob, json, re, sys


def collect(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void |frcnn::|\(.*$", "", r["Kernel_Name"])
        acc[name].append(float(r["Counter_Value"]) * 1024.0)
    return acc


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = {"method": __doc__.split("usage")[0].strip() + " Passes ran over `python bench.py --steps 3 --warmup 1 --no-cpu-baseline`.",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fv, wv = fetch.get(k, []), write.get(k, [])
    out["kernels"][k] = dict(launches=max(len(fv), len(wv)),
                             fetch_bytes_per_launch_corrected=2.0 * sum(fv) / max(len(fv), 1),
                             write_bytes_per_launch=sum(wv) / max(len(wv), 1))
for key, prefix in (("conv_igemm_k3", "conv_igemm_kernel<3, 8"), ("conv_x3", "conv_x3_kernel<3"), ("conv_wgradx", "conv_wgradx_kernel")):
    ks = [k for k in out["kernels"] if k.startswith(prefix)]
    n = sum(out["kernels"][k]["launches"] for k in ks)
    if n:
        out[key + "_bytes_per_launch"] = round(sum(out["kernels"][k]["launches"] * (out["kernels"][k]["fetch_bytes_per_launch_corrected"] + out["kernels"][k]["write_bytes_per_launch"]) for k in ks) / n)
rm = next((v for k, v in out["kernels"].items() if k.startswith("rmsprop_kernel")), None)
if rm:
    out["calibration_rmsprop"] = dict(fetch_corrected=rm["fetch_bytes_per_launch_corrected"], write=rm["write_bytes_per_launch"])
# provenance: bench.py copies it into roofline.traffic_taken, so that a table from another tree is visible in the bench line.
# (the GPU box has no .git: tools/refresh_round.sh passes the hash of the tree it was started from in FRCNN_GIT_HASH)
import datetime, os, subprocess
git = os.environ.get("FRCNN_GIT_HASH")
if not git:
    try:
        git = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
    except Exception:
        git = "unknown (no .git on the GPU box: set FRCNN_GIT_HASH)"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import csrc_sha256   # the hash bench.py compares before it prints roofline.traffic
out["taken"] = dict(git=git, date=datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), csrc_sha256=csrc_sha256())
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("bytes/launch: conv_x3", out.get("conv_x3_bytes_per_launch"), " conv_wgradx", out.get("conv_wgradx_bytes_per_launch"),
      " conv_igemm_k3", out.get("conv_igemm_k3_bytes_per_launch"), " rmsprop:", out.get("calibration_rmsprop"))
