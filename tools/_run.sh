python tools/_dbg.py 2>&1 | tail -6
