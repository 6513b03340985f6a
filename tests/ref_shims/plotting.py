"""Test-only headless stub shadowing the reference's utils/plotting.py (needs LaTeX)."""
import matplotlib
matplotlib.use("Agg")
import matplotlib.pyplot as plt


def newfig(width, nplots=1):
    fig = plt.figure()
    ax = fig.add_subplot(111)
    return fig, ax


def savefig(filename, crop=True):
    pass


def saveResultDir(save_path, save_hp):
    pass
