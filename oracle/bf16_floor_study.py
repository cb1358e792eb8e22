#!/usr/bin/env python3
"""How far apart are two SAMPLES of PyTorch's own mixed-precision error?  (VERDICT r5 #6b)

TEST INFRASTRUCTURE ONLY; executes the unmodified reference from /root/reference on CPU (build container, never the GPU box).

tests/test_gpu_baseline_shapes.py bounds the product's bf16 gradient error per tensor by max(5e-2, BF16_FLOOR_FACTOR x ebf[name]),
where ebf[name] is the relative L2 error of the reference under torch.autocast(cpu, bfloat16) against its fp64 self -- ONE sample of
that noise, taken with one summation order.  The product's error is another sample of the same kind of noise with another summation
order.  The factor must therefore cover the ratio of two independent samples, and it should be derived from a measurement of exactly
that, once -- not moved when a test fails.  This script measures it on the reference itself:

  sample A  the fixture's: autocast(bf16), batch as given                       (= ebf in tests/golden/<case>.npz, re-derived here)
  sample B  the same model and batch with the utterances in REVERSED order      (every weight-gradient sum runs over the rows
            (b, t) in another order; results are compared per parameter, which does not depend on the batch order)
  sample C  sample A's order on ONE intra-op thread                            (another blocking of every GEMM's reduction)

and reports, per parameter tensor (noise-driven ones excluded as in the test), the ratios e_B / e_A, e_C / e_A, e_B / e_C and their
inverses; the histogram and the quantiles over all tensors x ratios go to profiles/r06_bf16_floor_study.json.

usage: python oracle/bf16_floor_study.py <case> [<case> ...]      case in cfg0 | cfg1_b2 | cfg3_shape     (one subprocess per case)
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "profiles", "r06_bf16_floor_study.json")


def run(name):
    import numpy as np
    sys.path.insert(0, HERE)
    import gen_golden as G
    cfg = G.BIG[name]
    constant = G._boot(cfg["flags"])
    import torch
    from utils.metrics import calculate_metrics
    l2i, i2l = G.synth_labels(cfg["V"], constant)
    src, src_len, tgt = G.synth_batch(cfg["B"], cfg["T"], cfg["V"], cfg["src_len"], cfg["tgt_len"])
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

    def grads(mode, order=None, threads=None):
        if threads:
            torch.set_num_threads(threads)
        m = G.build_reference_model(constant, cfg, l2i, i2l)
        m.train()
        G.perturb_1d(m)
        s, sl, tg = src, src_len, tgt
        if order is not None:
            s, tg = src[order], tgt[order]
            sl = [src_len[i] for i in order] if isinstance(src_len, (list, tuple)) else src_len[order]
        if mode == "f64":
            m, s = m.double(), s.double()
            pr, go, _, _ = m(s, sl, tg)
            lo, _ = calculate_metrics(pr, go, smoothing=cfg["smoothing"], loss_type="ce")
        else:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                pr, go, _, _ = m(s, sl, tg)
                lo, _ = calculate_metrics(pr.float(), go, smoothing=cfg["smoothing"], loss_type="ce")
        lo.backward()
        return {k: q.grad.detach().double().numpy() for k, q in m.named_parameters()}

    nthr = torch.get_num_threads()
    g64 = grads("f64")
    rev = list(range(cfg["B"] - 1, -1, -1))
    gA = grads("bf16")
    gB = grads("bf16", order=rev)
    gC = grads("bf16", threads=1)
    torch.set_num_threads(nthr)
    emb = name.startswith("cfg3")
    noise = lambda k: k.endswith("key_linear.bias") or (emb and k in ("conv.0.bias", "conv.3.bias"))
    per = {}
    for k in g64:
        if noise(k):
            continue
        per[k] = {"eA": rel(gA[k], g64[k]), "eB": rel(gB[k], g64[k]), "eC": rel(gC[k], g64[k])}
    return {"case": name, "threads": nthr, "tensors": per}


def main():
    import numpy as np
    names = sys.argv[1:]
    if len(names) == 1 and names[0].startswith("@"):
        print(json.dumps(run(names[0][1:])))
        return
    results = []
    for n in names:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "@" + n], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("case %s failed:\n%s" % (n, r.stderr[-3000:]))
        results.append(json.loads(r.stdout.strip().splitlines()[-1]))
    ratios = []
    for r in results:
        for k, e in r["tensors"].items():
            for a, b in (("eA", "eB"), ("eA", "eC"), ("eB", "eC")):
                if min(e[a], e[b]) > 0:
                    ratios.append(max(e[a], e[b]) / min(e[a], e[b]))        # the larger sample over the smaller: what a one-sample floor must cover
    ratios = np.array(sorted(ratios))
    edges = [1.0, 1.05, 1.1, 1.2, 1.3, 1.5, 1.75, 2.0, 2.5, 3.0, 1e9]
    hist = {"[%.2f, %s)" % (edges[i], ("%.2f" % edges[i + 1]) if edges[i + 1] < 1e8 else "inf"): int(((ratios >= edges[i]) & (ratios < edges[i + 1])).sum())
            for i in range(len(edges) - 1)}
    q = lambda p: float(np.quantile(ratios, p))
    out = {"what": "ratio (larger / smaller) of two independent samples of the reference's OWN torch.autocast(cpu, bf16) gradient error against its "
                   "fp64 self, per parameter tensor: batch order reversed, and one intra-op thread, against the fixture's sample (oracle/bf16_floor_study.py)",
           "cases": [r["case"] for r in results], "n_ratios": int(len(ratios)), "histogram": hist,
           "quantiles": {"p50": q(0.5), "p90": q(0.9), "p99": q(0.99), "p99.9": q(0.999), "max": float(ratios.max())},
           "per_case": results}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("cases", "n_ratios", "histogram", "quantiles")}, indent=1))


if __name__ == "__main__":
    main()
