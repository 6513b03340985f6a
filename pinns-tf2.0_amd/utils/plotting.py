"""Headless figure helpers + result persistence.

Same entry points as the reference's utils/plotting.py (`newfig`, `savefig`,
`saveResultDir`, :8-16, :60-91) but with the Agg backend and no LaTeX/pgf dependency,
so the drop-in scripts finish on a GPU box without a TeX install.  `saveResultDir`
keeps the reference's layout results/<timestamp>-<script>/{graph.*, hp.json}; callers
may additionally hand it the trained flat weight vector, which the reference never
persisted (it has no checkpointing).
"""
import json
import os
import sys
from datetime import datetime

import matplotlib
matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402
import numpy as np  # noqa: E402


def _figsize(scale, nplots=1):
    width_in = 390.0 / 72.27 * scale
    return [width_in, nplots * width_in * (np.sqrt(5.0) - 1.0) / 2.0]


def newfig(width, nplots=1):
    fig = plt.figure(figsize=_figsize(width, nplots))
    return fig, fig.add_subplot(111)


def savefig(filename, crop=True):
    kw = {"bbox_inches": "tight", "pad_inches": 0} if crop else {}
    plt.savefig("{}.pdf".format(filename), **kw)
    plt.savefig("{}.png".format(filename), **kw)


def is_root():
    """rank 0 of a data-parallel launch (or a plain run): the one process that writes results/"""
    return int(os.environ.get("WORLD_SIZE", "1")) <= 1 or int(os.environ.get("RANK", "0")) == 0


def saveResultDir(save_path, save_hp, weights=None):
    if not is_root():
        return None
    stamp = datetime.now().strftime("%Y%m%d-%H%M%S")
    script = os.path.splitext(os.path.basename(sys.argv[0]))[0]
    res_dir = os.path.join(save_path, "results", "{}-{}".format(stamp, script))
    os.makedirs(res_dir, exist_ok=True)
    print("Saving results to directory ", res_dir)
    savefig(os.path.join(res_dir, "graph"))
    with open(os.path.join(res_dir, "hp.json"), "w") as f:
        json.dump(save_hp, f)
    if weights is not None:
        np.save(os.path.join(res_dir, "weights.npy"), np.asarray(weights, dtype=np.float64))
    return res_dir
