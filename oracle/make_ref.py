#!/usr/bin/env python3
"""Build the artefact of the Tier-A CPU baseline ("the reference's scripts over the torch-CPU shim, timed
on the GPU host").

TEST INFRASTRUCTURE.  The reference is pure Python, so there is nothing to compile: the "binary" of this oracle is
oracle/_ref/reference_sources.tar.gz, an archive of the few files the Burgers inference path is made of, packed from
where they lie under /root/reference.  oracle/_ref/ is git-ignored (no reference source ever enters the history or
sits in the tree as a source file) but travels to the GPU box with the snapshot, like a built .so;
oracle/ref_baseline.py unpacks it into a temporary directory for the duration of one timing run.  Run by
__graft_entry__.build() whenever /root/reference is present.

    python3 oracle/make_ref.py            # -> oracle/_ref/reference_sources.tar.gz
"""
import io
import os
import sys
import tarfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
DST = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(DST, "reference_sources.tar.gz")
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
    os.makedirs(DST, exist_ok=True)
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w:gz") as tar:
        for rel in FILES:
            tar.add(os.path.join(REF, rel), arcname=rel)
    with open(ARCHIVE + ".tmp", "wb") as fh:
        fh.write(buf.getvalue())
    os.replace(ARCHIVE + ".tmp", ARCHIVE)
    for rel in FILES:                    # earlier layout of this directory: plain copies -- remove them
        old = os.path.join(DST, rel)
        if os.path.exists(old):
            os.remove(old)
    if verbose:
        print("packed %d reference files into %s" % (len(FILES), ARCHIVE))
    return True


def staged():
    return os.path.exists(ARCHIVE)


def unpack(dst):
    """extract the archive under dst (a scratch directory the caller owns and deletes)"""
    with tarfile.open(ARCHIVE, mode="r:gz") as tar:
        names = tar.getnames()
        assert sorted(names) == sorted(FILES), names
        tar.extractall(dst)
    return dst


if __name__ == "__main__":
    sys.exit(0 if stage() else 1)
