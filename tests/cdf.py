"""Independent reader for the netCDF classic formats CDF-1/2/5 (tests only) -- written from the format specification, not from
the writer in miniweatherml_amd/csrc/mw_netcdf.cpp, so that the two check each other.  Returns dims, variables (as numpy arrays,
record variables with the record axis first) and the raw layout (begin offsets, vsize)."""
import struct

import numpy as np

TYPES = {1: ("i1", 1), 2: ("S1", 1), 3: (">i2", 2), 4: (">i4", 4), 5: (">f4", 4), 6: (">f8", 8)}


class Reader:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        self.pos = 0
        if self.b[:3] != b"CDF" or self.b[3] not in (1, 2, 5):
            raise ValueError("not a netCDF classic file")
        self.version = self.b[3]
        self.pos = 4
        self.numrecs = self.nonneg()
        self.dims = self.dim_list()
        self.gatts = self.att_list()
        self.vars = self.var_list()

    def i32(self):
        v = struct.unpack(">i", self.b[self.pos:self.pos + 4])[0]
        self.pos += 4
        return v

    def i64(self):
        v = struct.unpack(">q", self.b[self.pos:self.pos + 8])[0]
        self.pos += 8
        return v

    def nonneg(self):
        return self.i64() if self.version == 5 else self.i32()

    def name(self):
        n = self.nonneg()
        s = self.b[self.pos:self.pos + n].decode()
        self.pos += (n + 3) // 4 * 4
        return s

    def dim_list(self):
        tag, n = self.i32(), self.nonneg()
        assert (tag, n) == (0, 0) or tag == 10, "bad dim_list tag %d" % tag
        return [(self.name(), self.nonneg()) for _ in range(n)]

    def att_list(self):
        tag, n = self.i32(), self.nonneg()
        assert (tag, n) == (0, 0) or tag == 12, "bad att_list tag %d" % tag
        out = []
        for _ in range(n):
            nm, ty, ne = self.name(), self.i32(), self.nonneg()
            nbytes = ne * TYPES[ty][1]
            out.append((nm, ty, self.b[self.pos:self.pos + nbytes]))
            self.pos += (nbytes + 3) // 4 * 4
        return out

    def var_list(self):
        tag, n = self.i32(), self.nonneg()
        assert (tag, n) == (0, 0) or tag == 11, "bad var_list tag %d" % tag
        out = []
        for _ in range(n):
            nm = self.name()
            nd = self.nonneg()
            dimids = [self.nonneg() for _ in range(nd)]
            atts = self.att_list()
            ty = self.i32()
            vsize = self.nonneg()
            begin = self.i64() if self.version >= 2 else self.i32()
            out.append(dict(name=nm, dimids=dimids, atts=atts, type=ty, vsize=vsize, begin=begin))
        self.header_bytes = self.pos
        return out

    def is_rec(self, v):
        return bool(v["dimids"]) and self.dims[v["dimids"][0]][1] == 0

    def recsize(self):
        return sum(v["vsize"] for v in self.vars if self.is_rec(v))

    def get(self, name):
        v = next(x for x in self.vars if x["name"] == name)
        dt, sz = TYPES[v["type"]]
        shape = [self.dims[d][1] for d in v["dimids"]]
        if not self.is_rec(v):
            n = int(np.prod(shape)) if shape else 1
            return np.frombuffer(self.b, dtype=dt, count=n, offset=v["begin"]).reshape(shape).astype(dt[1:] if dt[0] == ">" else dt)
        inner = shape[1:]
        n = int(np.prod(inner)) if inner else 1
        rs = self.recsize()
        recs = [np.frombuffer(self.b, dtype=dt, count=n, offset=v["begin"] + r * rs).reshape(inner) for r in range(self.numrecs)]
        return np.array(recs).astype(dt[1:] if dt[0] == ">" else dt).reshape([self.numrecs] + inner)
