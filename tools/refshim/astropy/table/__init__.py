class Table:
    @staticmethod
    def read(*a, **k):
        raise RuntimeError('astropy.table.Table is a placeholder in refshim')
