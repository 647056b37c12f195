#!/usr/bin/env python3
"""Drop-in for the reference's rank.py command line (rank.py:129-391), on the MI355X engine."""
import eps_amd  # noqa: F401  (registers the package)
from eps_amd.rank_stage import main

if __name__ == "__main__":
    main()
