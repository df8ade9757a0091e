"""Prints the headline fields of a bench.py result line: show_line.py <file>."""
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        print({k: d.get(k) for k in ("value", "n_gpus", "ranks", "processes", "ms_per_step", "gathered_frame_equals_single_gpu_frame", "rehearsal", "transport")},
              d["config"].get("multi_gpu_path"), "in_library" in d and d["in_library"])
