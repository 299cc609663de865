"""Builders for the data-driven cross-table-lookup description (include/zkm_hip.h: zkm_column, zkm_colset,
zkm_ctl_table, zkm_ctl_z, zkm_cross_table_lookup).  Mirrors the reference's Column / Filter / TableWithColumns /
CrossTableLookup constructors (prover/src/cross_table_lookup.rs:31-415) so table descriptions read like
all_stark.rs.  Pure host-side data marshalling: no computation happens here.
"""
import ctypes as C

import numpy as np

COLUMN_DT = np.dtype([("n_local", "u4"), ("n_next", "u4"), ("term_off", "u4"), ("_pad", "u4"), ("constant", "u8")])
COLSET_DT = np.dtype([("ncols", "u4"), ("col_off", "u4"), ("has_filter", "u4"), ("nprod", "u4"), ("prod_off", "u4"),
                      ("nconst", "u4"), ("const_off", "u4"), ("_pad", "u4")])
CTLZ_DT = np.dtype([("ncolsets", "u4"), ("colset_off", "u4"), ("num_helpers", "u4"), ("_pad", "u4"), ("beta", "u8"), ("gamma", "u8")])
SIDE_DT = np.dtype([("table", "u4"), ("colset", "u4")])
CTL_DT = np.dtype([("nlooking", "u4"), ("looking_off", "u4"), ("looked_table", "u4"), ("looked_colset", "u4")])


class CtlTableStruct(C.Structure):
    _fields_ = [("columns", C.c_void_p), ("ncolumns", C.c_size_t), ("term_col", C.c_void_p), ("term_coeff", C.c_void_p),
                ("nterms", C.c_size_t), ("colsets", C.c_void_p), ("ncolsets", C.c_size_t), ("filter_idx", C.c_void_p),
                ("nfilter_idx", C.c_size_t)]


class TableInputStruct(C.Structure):
    _fields_ = [("table_id", C.c_int), ("trace", C.c_void_p), ("ncols", C.c_size_t), ("log_n", C.c_uint), ("ctl", C.c_void_p),
                ("columns", C.c_void_p)]


class CtlTable:
    """Column sets of one table.  column()/single()/... return column indices; colset() returns a column-set index."""

    def __init__(self):
        self._cols, self._tc, self._tf, self._sets, self._fidx = [], [], [], [], []
        self._packed = None

    # ---- Column constructors (cross_table_lookup.rs:128-245)
    def column(self, local=(), next=(), constant=0):
        off = len(self._tc)
        for c, f in list(local) + list(next):
            self._tc.append(int(c))
            self._tf.append(int(f))
        self._cols.append((len(local), len(next), off, 0, int(constant)))
        self._packed = None
        return len(self._cols) - 1

    def single(self, c):
        return self.column(local=[(c, 1)])

    def constant(self, v):
        return self.column(constant=v)

    def le_bits(self, cs):
        return self.column(local=[(c, 1 << i) for i, c in enumerate(cs)])

    def le_bytes(self, cs):
        return self.column(local=[(c, 1 << (8 * i)) for i, c in enumerate(cs)])

    def sum(self, cs):
        return self.column(local=[(c, 1) for c in cs])

    # ---- TableWithColumns (columns must be created consecutively: they are referenced as a range)
    def colset(self, columns, filter_products=None, filter_constants=None):
        columns = list(columns)
        assert columns == list(range(columns[0], columns[0] + len(columns))), "columns of a set must be consecutive indices"
        has_filter = filter_products is not None or filter_constants is not None
        prods = list(filter_products or [])
        consts = list(filter_constants or [])
        poff = len(self._fidx)
        for a, b in prods:
            self._fidx += [int(a), int(b)]
        coff = len(self._fidx)
        self._fidx += [int(c) for c in consts]
        self._sets.append((len(columns), columns[0], int(has_filter), len(prods), poff, len(consts), coff, 0))
        self._packed = None
        return len(self._sets) - 1

    def singles_set(self, cols, filter_col=None):
        """Column::singles(cols) with Filter::new_simple(Column::single(filter_col))."""
        first = len(self._cols)
        for c in cols:
            self.single(c)
        f = None if filter_col is None else [self.single(filter_col)]
        return self.colset(range(first, first + len(cols)), filter_constants=f)

    def pack(self):
        if self._packed is None:
            cols = np.array(self._cols, dtype=COLUMN_DT) if self._cols else np.zeros(0, dtype=COLUMN_DT)
            tc = np.array(self._tc, dtype=np.uint32)
            tf = np.array(self._tf, dtype=np.uint64)
            sets = np.array(self._sets, dtype=COLSET_DT) if self._sets else np.zeros(0, dtype=COLSET_DT)
            fidx = np.array(self._fidx, dtype=np.uint32)
            st = CtlTableStruct(cols.ctypes.data, len(cols), tc.ctypes.data, tf.ctypes.data, len(tc), sets.ctypes.data, len(sets),
                                fidx.ctypes.data, len(fidx))
            self._packed = (st, cols, tc, tf, sets, fidx)
        return self._packed[0]


def make_zs(entries):
    """entries: list of (colset_ids, beta, gamma[, num_helpers]).  Returns (zs array, colset_ids array)."""
    zs = np.zeros(len(entries), dtype=CTLZ_DT)
    ids = []
    for i, e in enumerate(entries):
        cs, beta, gamma = e[0], e[1], e[2]
        nh = e[3] if len(e) > 3 else ((len(cs) + 1) // 2 if len(cs) > 1 else 0)
        zs[i] = (len(cs), len(ids), nh, 0, beta, gamma)
        ids += list(cs)
    return zs, np.array(ids, dtype=np.uint32)


def pack_ctls(ctls):
    """ctls: list of (looking=[(table, colset), ...], looked=(table, colset))."""
    sides, arr = [], np.zeros(len(ctls), dtype=CTL_DT)
    for i, (looking, looked) in enumerate(ctls):
        arr[i] = (len(looking), len(sides), looked[0], looked[1])
        sides += list(looking)
    return arr, np.array(sides, dtype=SIDE_DT) if sides else np.zeros(0, dtype=SIDE_DT)


def pack_tables(tables):
    """tables: list of (table_id, trace_ptr_int, ncols, log_n, CtlTable); trace_ptr_int may be a LIST of ncols column pointers
    (zkm_table_input.columns: the reference's Vec<PolynomialValues>).  Returns (ctypes array, keepalive)."""
    arr = (TableInputStruct * len(tables))()
    keep = []
    for i, (tid, ptr, ncols, log_n, ctl) in enumerate(tables):
        st = ctl.pack()
        keep.append(st)
        if isinstance(ptr, (list, tuple)):
            assert len(ptr) == ncols
            cols = (C.c_void_p * ncols)(*ptr)
            keep.append(cols)
            arr[i] = TableInputStruct(tid, None, ncols, log_n, C.addressof(st), C.addressof(cols))
        else:
            arr[i] = TableInputStruct(tid, ptr, ncols, log_n, C.addressof(st), None)
    return arr, keep


def derive_zs(ntables, ctls, challenges):
    """The CtlZData of every table in the order cross_table_lookup_data builds them (cross_table_lookup.rs:634-703): for each lookup,
    for each challenge (beta, gamma): one Z per run of consecutive looking entries of the same table (group_by, :807; its helper
    columns pair the run's column sets, :474-481), then one Z for the looked table.  ctls as pack_ctls takes them; challenges = flat
    [beta0, gamma0, beta1, gamma1, ...].  Returns [(zs array (CTLZ_DT), colset_ids array)] per table -- the host-side twin of
    csrc/ctl.hip derive_zs, for tests that want a table's CtlData on its own (Context.ctl_data / Oracle.ctl_data)."""
    entries = [[] for _ in range(ntables)]
    nch = len(challenges) // 2
    for looking, looked in ctls:
        for ch in range(nch):
            beta, gamma = int(challenges[2 * ch]), int(challenges[2 * ch + 1])
            i = 0
            while i < len(looking):
                j = i
                while j < len(looking) and looking[j][0] == looking[i][0]:
                    j += 1
                entries[looking[i][0]].append(([cs for _, cs in looking[i:j]], beta, gamma))
                i = j
            entries[looked[0]].append(([looked[1]], beta, gamma))
    return [make_zs(e) if e else (np.zeros(0, dtype=CTLZ_DT), np.zeros(0, dtype=np.uint32)) for e in entries]
