"""Calibration and large-sample audit scripts of the parity tests (run by hand on a GPU box, not collected by pytest).  They
live under tests/ because, like the tests, they use the CPU oracle as the checker."""
