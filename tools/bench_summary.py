"""One line of the figures quoted in README.md from a bench.py JSON line:  python tools/bench_summary.py gpurun_out/bench.json"""
import json, sys
d = json.load(open(sys.argv[1]))
o = d.get("other_configs") or {}
print("train tok/s", d["value"], "ms/step", d["ms_per_step"], "| decode tok/s", d["decode"]["tokens_per_sec"], "ms/token-step",
      d["decode"]["ms_per_token_step"], "hbm frac", d["decode"]["roofline"]["frac"], "| gemm frac", d["roofline"]["frac"], "block frac",
      d["roofline"]["block_forward"]["frac"], "| configs[3] ms/step", (o.get("configs[3]") or {}).get("ms_per_step"), "| configs[4] ms/token",
      (o.get("configs[4]") or {}).get("ms_per_token"), "TB/s", (o.get("configs[4]") or {}).get("weight_stream_TBps"))
