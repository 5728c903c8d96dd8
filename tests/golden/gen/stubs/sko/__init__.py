"""Import shim (ours): scikit-opt is only used by the reference's unused `maven` heuristic (EA_compare.py)."""
