#!/usr/bin/env python3
"""Stage the reference's own sources for the Tier-A CPU baseline ("the reference's scripts over the
torch-CPU shim, timed on the GPU host").

TEST INFRASTRUCTURE.  The reference is pure Python, so there is nothing to compile: "building" oracle/_ref means
copying the few files the Burgers inference path is made of, from where they lie under /root/reference, into
oracle/_ref/ -- which is git-ignored (no reference source ever enters the history) but travels to the GPU box with
the snapshot, like a built .so.  Run by __graft_entry__.build() whenever /root/reference is present.

    python3 oracle/make_ref.py            # -> oracle/_ref/{utils,1d-burgers}/...
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
DST = os.path.join(HERE, "_ref")
FILES = [
    "utils/neuralnetwork.py",            # NeuralNetwork: model, grad, Adam loop, flat layout, fit (hot path rows 1-11)
    "utils/custom_lbfgs.py",             # lbfgs, Struct
    "utils/logger.py",                   # Logger
    "1d-burgers/inf_cont_burgers.py",    # BurgersInformedNN.f_model / loss + the driver
    "1d-burgers/burgersutil.py",         # prep_data
    "1d-burgers/data/burgers_shock.mat",
]


def stage(verbose=True):
    if not os.path.isdir(REF):
        return False
    for rel in FILES:
        dst = os.path.join(DST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
    if verbose:
        print("staged %d reference files under %s" % (len(FILES), DST))
    return True


def staged():
    return all(os.path.exists(os.path.join(DST, rel)) for rel in FILES)


if __name__ == "__main__":
    sys.exit(0 if stage() else 1)
