#!/usr/bin/env python3
"""python tools/run_with_watchdog.py SECONDS script.py [args]: run a script and dump every thread's Python stack if it is still
running after SECONDS (then exit) -- to see WHERE a profiled run stopped making progress."""
import faulthandler
import runpy
import sys

secs = float(sys.argv[1])
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
