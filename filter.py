#!/usr/bin/env python3
"""Drop-in for the reference's filter.py command line (filter.py:26-166), on the MI355X engine."""
import eps_amd  # noqa: F401  (registers the package)
from eps_amd.filter_stage import main

if __name__ == "__main__":
    main()
