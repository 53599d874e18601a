"""Thin stdout / text-file logger with the two calls the sampling CLI uses (configure, log, get_dir).
The reference's OpenAI-baselines logger (json/csv/tensorboard/wandb, MPI-averaged scalars) is training
infrastructure and out of scope."""
import datetime
import os
import sys

_DIR = None
_FILE = None


def configure(args=None, comm=None, dir=None):
    """Log directory: loggings/<args.dir> like the reference (logger.py:458-497)."""
    global _DIR, _FILE
    sub = dir if dir is not None else getattr(args, "dir", "") or datetime.datetime.now().strftime("rgm-%Y-%m-%d-%H-%M-%S")
    _DIR = os.path.join("loggings", sub)
    os.makedirs(_DIR, exist_ok=True)
    if int(os.environ.get("RANK", "0")) == 0:
        _FILE = open(os.path.join(_DIR, "log.txt"), "a")
    log(f"Logging to {_DIR}")


def get_dir():
    return _DIR


def log(*args):
    msg = " ".join(str(a) for a in args)
    if int(os.environ.get("RANK", "0")) == 0:
        print(msg, file=sys.stdout, flush=True)
        if _FILE is not None:
            _FILE.write(msg + "\n")
            _FILE.flush()
