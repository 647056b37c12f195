"""Per-run Hits bookkeeping for the rank stage: one (train, valid, test) triple per evaluation, a summary per run and the
mean +- std over runs, printed in the line format rank.py's log scrapers expect (role of the reference's logger.py)."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

_LABELS = ("Highest Train", "Highest Valid", "  Final Train", "   Final Test")


class Logger:
    def __init__(self, runs: int, info=None):
        self.info = info
        self.results: List[List[Tuple[float, float, float]]] = [[] for _ in range(runs)]

    def add_result(self, run: int, result) -> None:
        if len(result) != 3 or not 0 <= run < len(self.results):
            raise ValueError("expected a (train, valid, test) triple for an existing run")
        self.results[run].append(tuple(float(x) for x in result))

    def summary(self, run: int) -> Optional[Dict[str, float]]:
        """Best train / best valid over the run's evaluations, and train / test AT the best-valid evaluation
        (percent).  None when the run has no evaluation yet."""
        if not self.results[run]:
            return None
        t = 100.0 * torch.tensor(self.results[run], dtype=torch.float64)
        at = int(torch.argmax(t[:, 1]))
        return {_LABELS[0]: float(t[:, 0].max()), _LABELS[1]: float(t[:, 1].max()), _LABELS[2]: float(t[at, 0]),
                _LABELS[3]: float(t[at, 2])}

    def print_statistics(self, run: Optional[int] = None) -> None:
        if run is not None:
            print(f"Run {run + 1:02d}:")
            for label, value in (self.summary(run) or {}).items():
                print(f"{label}: {value:.2f}")
            return
        per_run = [s for s in (self.summary(r) for r in range(len(self.results))) if s is not None]
        print("All runs:")
        for label in _LABELS:
            col = torch.tensor([s[label] for s in per_run], dtype=torch.float64)
            spread = float(col.std()) if col.numel() > 1 else 0.0
            print(f"{label}: {float(col.mean()):.2f} ± {spread:.2f}")
