#!/usr/bin/env python3
"""usage: run_env.py VAR=value [VAR=value ...] script.py [args...] -- runs the script IN THIS PROCESS with the variables
set (rocprofv3 must be handed the program itself: an `env` or shell hop in front of a process that has initialised the
GPU is refused on the GPU boxes)."""
import os
import runpy
import sys

args = sys.argv[1:]
while args and "=" in args[0] and not args[0].endswith(".py"):
    k, v = args.pop(0).split("=", 1)
    os.environ[k] = v
sys.argv = args
runpy.run_path(args[0], run_name="__main__")
