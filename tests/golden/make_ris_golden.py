"""Generate tests/golden/ris_notebook.npz from the reference's own notebook.

Run in the BUILD container only (needs /root/reference):
    python tests/golden/make_ris_golden.py
It executes the code cells of restir_di/RIS_Test/ris_test.ipynb that define the proposal pairs and
sampleRIS (cells 2/3 and 5; raw file lines 28-104 and 172-188), with np.random seeded and
np.random.choice wrapped so the resampled indices are recorded, and stores inputs + outputs for
K samples per configuration.  The committed .npz is data only (no reference source).
"""
import json
import os

import numpy as np

REF = "/root/reference/restir_di/RIS_Test/ris_test.ipynb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ris_notebook.npz")
K = 256
SEED = 12345


def main():
    nb = json.load(open(REF))
    cells = ["".join(c["source"]) for c in nb["cells"] if c["cell_type"] == "code"]

    def strip_plots(src):
        keep = []
        for line in src.splitlines():
            if any(tok in line for tok in ("figure", "fig_", "plt.", "xs = np.linspace", "p1s =", "p2s =")):
                continue
            keep.append(line)
        return "\n".join(keep)

    import matplotlib
    matplotlib.use("Agg")
    out = {}
    for pair, defs_cell in (("A", cells[1]), ("B", cells[4])):
        ns = {}
        exec("import numpy as np\nimport math", ns)
        exec(strip_plots(defs_cell), ns)
        exec(cells[2], ns)  # sampleRIS
        for M in (2, 4, 10, 20):
            np.random.seed(SEED + M)
            picks = []
            orig_choice = np.random.choice

            def recording_choice(a, p=None, _orig=orig_choice, _picks=picks):
                r = _orig(a, p=p)
                _picks.append(int(r))
                return r

            np.random.choice = recording_choice
            try:
                us = np.random.rand(M, K)
                samples, biased_ws, naive_ws, mis_ws = ns["sampleRIS"](us)
            finally:
                np.random.choice = orig_choice
            # np.apply_along_axis probes the function once on the first column before the loop
            picks = picks[-K:]
            key = f"{pair}_M{M}_"
            out[key + "us"] = us
            out[key + "indices"] = np.array(picks, np.int64)
            out[key + "samples"] = samples
            out[key + "biased_ws"] = biased_ws
            out[key + "naive_ws"] = naive_ws
            out[key + "mis_ws"] = mis_ws
    # statistical anchors of the notebook run at N = 50 000 (seeded), E[f(x) * W]
    ns = {}
    exec("import numpy as np\nimport math", ns)
    exec(strip_plots(cells[1]), ns)
    exec(cells[2], ns)
    np.random.seed(SEED)
    for M in (2, 4, 10, 20):
        us = np.random.rand(M, 50000)
        xs, b, n, m = ns["sampleRIS"](us)
        f = ns["target_density"](xs)
        out[f"mean_M{M}"] = np.array([np.mean(f * b), np.mean(f * n), np.mean(f * m)])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.startswith("mean")})
    for M in (2, 4, 10, 20):
        print(M, out[f"mean_M{M}"])


if __name__ == "__main__":
    main()
