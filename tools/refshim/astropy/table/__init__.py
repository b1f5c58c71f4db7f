"""astropy.table stand-in: only ``Table.read`` of a small CSV file (header line + numeric rows), which is how the
reference loads its DESI instrumental-systematics table (reference vega/correlation_func.py:586-591)."""
import numpy as np


class Table(dict):
    @staticmethod
    def read(path, *args, **kwargs):
        with open(str(path)) as f:
            names = f.readline().strip().split(',')
        data = np.atleast_2d(np.loadtxt(str(path), delimiter=',', skiprows=1))
        return Table({name: data[:, i] for i, name in enumerate(names)})
