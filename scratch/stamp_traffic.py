"""Copy a PMC capture's hbm_traffic.json (gpurun_out/prof_<tag>/, written by scratch/profile_round.sh on the GPU box) into
profiles/, stamped with the commit whose kernel sources it was captured from (the GPU box has no .git).  Refuses a capture whose
kernel-source hash is not the working tree's.   usage: python scratch/stamp_traffic.py r06"""
import json, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import kernel_sources_sha, KERNEL_SOURCES
tag = sys.argv[1]
rec = json.loads((ROOT / "gpurun_out" / f"prof_{tag}" / "hbm_traffic.json").read_text())
if rec.get("_kernel_sources_sha") != kernel_sources_sha():
    sys.exit("capture was taken from other kernel sources than this tree's")
paths = [f"scdeepsort_amd/csrc/{n}" for n in KERNEL_SOURCES]
dirty = subprocess.run(["git", "status", "--porcelain", "--", *paths], cwd=ROOT, capture_output=True, text=True).stdout.strip()
commit = subprocess.run(["git", "log", "-1", "--format=%H", "--", *paths], cwd=ROOT, capture_output=True, text=True).stdout.strip()
rec["_commit"] = commit + (" + uncommitted kernel edits" if dirty else " (last commit that touched the kernel sources)")
(ROOT / "profiles" / "hbm_traffic.json").write_text(json.dumps(rec, indent=1))
print("stamped", rec["_commit"], rec["_kernel_sources_sha"][:12], rec["_popularity"])
