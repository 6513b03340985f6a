"""Runs one of the drop-in scripts (pinns-tf2.0_amd/1d-burgers/inf_cont_burgers.py, .../inf_cont_schrodinger.py, ...)
as the process it is launched in -- a plain `python` run or one rank of
`python -m torch.distributed.run --nproc-per-node N tests/helpers/dp_script.py <script> <hp.json> <out_dir>` --
and records what tests/test_gpu_dp_scripts.py compares between the two:

  <out_dir>/rank<r>.json   every (tag, epoch, loss) the script handed to Logger.log_train_epoch at FULL precision (the
                           printed lines carry 4 digits), the final error, comm mode, L-BFGS restarts
  <out_dir>/rank<r>.npy    the trained flat weight vector of this rank's replica
  <out_dir>/rank<r>.out    this rank's stdout (rank 0: the reference-format log; other ranks: must be empty)

The script itself runs unmodified through runpy (its `if __name__ == "__main__"` body), exactly as the CLI would."""
import contextlib
import io
import json
import os
import runpy
import sys

import numpy as np


def main():
    script, hp_path, out_dir = sys.argv[1:4]
    rank = int(os.environ.get("RANK", "0"))
    pkg = os.path.dirname(os.path.dirname(os.path.abspath(script)))
    sys.path.insert(0, os.path.join(pkg, "utils"))
    sys.path.insert(0, pkg)
    import logger
    record = []
    inner = logger.Logger.log_train_epoch

    def log_train_epoch(self, epoch, loss, custom="", is_iter=False):
        record.append(["nt" if is_iter else "tf", int(epoch), float(loss)])
        return inner(self, epoch, loss, custom, is_iter)

    logger.Logger.log_train_epoch = log_train_epoch
    sys.argv = [script, hp_path]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g = runpy.run_path(script, run_name="__main__")
    pinn = g["pinn"]
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), pinn.get_weights())
    with open(os.path.join(out_dir, "rank%d.out" % rank), "w") as fh:
        fh.write(buf.getvalue())
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as fh:
        json.dump({"log": record, "comm_mode": pinn.comm_mode, "restarts": pinn.nt_restarts,
                   "error": float(pinn.logger.get_error_u()) if pinn.logger.error_fn else None,
                   "n_f_local": pinn._engine.n_f, "n_u_local": pinn._engine.n_u, "n_b_local": pinn._engine.n_b,
                   "device_evals": pinn.status()[0]}, fh)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:      # (a plain run never imports torch)
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
