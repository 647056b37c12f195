"""Per-run result table with mean +- std printing (the role of the reference's logger.py:9-44)."""
import torch


class Logger:
    def __init__(self, runs, info=None):
        self.info = info
        self.results = [[] for _ in range(runs)]

    def add_result(self, run, result):
        assert len(result) == 3 and 0 <= run < len(self.results)
        self.results[run].append(result)

    def print_statistics(self, run=None):
        if run is not None:
            r = 100 * torch.tensor(self.results[run])
            am = r[:, 1].argmax().item()
            print(f'Run {run + 1:02d}:')
            print(f'Highest Train: {r[:, 0].max():.2f}')
            print(f'Highest Valid: {r[:, 1].max():.2f}')
            print(f'  Final Train: {r[am, 0]:.2f}')
            print(f'   Final Test: {r[am, 2]:.2f}')
            return
        best = []
        for rr in self.results:
            if not rr:
                continue
            r = 100 * torch.tensor(rr)
            am = r[:, 1].argmax()
            best.append((r[:, 0].max().item(), r[:, 1].max().item(), r[am, 0].item(), r[am, 2].item()))
        b = torch.tensor(best)
        print('All runs:')
        for i, name in enumerate(['Highest Train', 'Highest Valid', '  Final Train', '   Final Test']):
            col = b[:, i]
            print(f'{name}: {col.mean():.2f} ± {col.std() if len(col) > 1 else 0.0:.2f}')
