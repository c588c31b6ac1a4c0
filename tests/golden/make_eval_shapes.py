"""The image sizes of the reference's own 500-image test set, in the order its eval loop met them, parsed out of the log the reference ships
(experiments/.../exp_0/logs/exp_debug.log.1: the per-image lines agents/llicti_agent.py:154-162 prints; the LAST complete run of 500).
Data only: (H, W) pairs.  bench.py's `api_path_mixed` leg and tests/test_hip_parity.py draw synthetic images of exactly these sizes
(the images themselves are not in the reference tree).  Run here (needs /root/reference); the output is committed."""
import collections
import glob
import json
import os
import re

REF = os.environ.get("LLICTI_REFERENCE", "/root/reference")


def main():
    logs = glob.glob(os.path.join(REF, "experiments", "*", "exp_0", "logs", "exp_debug.log.1"))
    assert len(logs) == 1, logs
    runs, cur = [], []
    for line in open(logs[0]):
        m = re.search(r"Agent - : +(\d+) +(\d+)x(\d+) +bpsp", line)
        if not m:
            continue
        i, h, w = map(int, m.groups())
        if i == 0 and cur:
            runs.append(cur)
            cur = []
        cur.append((i, h, w))
    if cur:
        runs.append(cur)
    full = [r for r in runs if len(r) == 500 and [i for i, _, _ in r] == list(range(500))]
    shapes = [[h, w] for _, h, w in full[-1]]
    cnt = collections.Counter(map(tuple, shapes))
    out = {"source": "reference eval log, last complete run of 500 images (sizes only)", "n": len(shapes), "distinct": len(cnt),
           "megapixels": sum(h * w for h, w in shapes) / 1e6,
           "most_common": [[list(k), v] for k, v in cnt.most_common(5)], "shapes": shapes}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval_shapes.json")
    with open(dst, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(dst, out["n"], out["distinct"], out["megapixels"], out["most_common"])


if __name__ == "__main__":
    main()
