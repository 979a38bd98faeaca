"""Puts BASELINE's other configs on record -> gpurun_out/r02_configs.json (copied to profiles/):
  cfg2 forward (bench.py --config cfg2), cfg3 full-graph training step = cfg4 at N = 1 (examples/train_sharded.py),
  cfg4 mini-batch epoch through the graphed step, cfg5 (764,741 cells, fp16-stored features) forward on ONE GPU."""
import json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
out = {"_how": "python scratch/configs_record.py on one MI355X (gpurun); every entry is the command's own output"}
def run(cmd, env=None, timeout=900):
    t = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    return r.stdout, r.stderr, round(time.time() - t, 1)
for cfg in ("cfg2", "cfg5"):
    so, se, dt = run([sys.executable, "bench.py", "--config", cfg, "--steps", "20", "--warmup", "3"] + (["--no-cpu-baseline"] if cfg == "cfg5" else []))
    line = [l for l in so.splitlines() if l.startswith("{")]
    d = json.loads(line[-1]) if line else {"error": se[-500:]}
    if "roofline" in d:
        d["roofline"].pop("note", None)
    out[cfg + "_forward_1gpu"] = {"cmd": f"python bench.py --config {cfg} --steps 20 --warmup 3", "wall_s": dt, "line": d}
so, se, dt = run([sys.executable, "examples/train_sharded.py", "--config", "cfg3", "--steps", "10"])
out["cfg4_full_batch_training_step_1gpu"] = {"cmd": "python examples/train_sharded.py --config cfg3 --steps 10", "wall_s": dt,
                                             "stdout": so.strip().splitlines()[-1] if so.strip() else se[-500:]}
Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "r02_configs.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1)[:3000])
