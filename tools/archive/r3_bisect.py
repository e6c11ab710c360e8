"""Round-3 opening diagnostic (GPU): why was the benched-path caption check red?

1. GPU vs GPU: the same real crops through an 8-row and a 128-row caption plan at 768x768 (tests/gpu_checks.py::check_plan_capacity)
   — with the calibrated stand-in captioner ("v2", tools/make_weights.py::CAPTION_STANDIN) and with the round-2 one ("v1").
2. GPU vs transformers-on-CPU on the same kind of crops, per stand-in.
One JSON object per line on stdout.  usage: python tools/r3_bisect.py [v2] [v1]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import gpu_checks as G
    for standin in (sys.argv[1:] or ["v2", "v1"]):
        out, cap = G.check_plan_capacity(R=768, n=16, small=8, large=128, standin=standin)
        print(json.dumps({"check": "plan_capacity", "standin": standin, **out}), flush=True)
        del cap
        torch.cuda.empty_cache()
        out, cap = G.check_captioner_real_crops(R=768, n=4, capacity=128, standin=standin)
        print(json.dumps({"check": "real_crops_vs_cpu", "standin": standin, **out}), flush=True)
        del cap
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
