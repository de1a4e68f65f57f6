"""Host-side mirror of miniWeatherML's `core::Coupler` (model/core/coupler.h:17-493) and
`core::DataManager` (model/core/DataManager.h:20-585) for Python callers.

Same names, argument meaning and error behaviour (endrun -> MWError) as the reference for the calls the
hot path needs: options, tracer registry, grid getters, decomposition, `register_and_allocate`, `get`,
`get_lev_col`, `get_collapsed`.  Storage is torch (plumbing only): every field is a contiguous fp64 CUDA
tensor whose data_ptr() goes straight into the C ABI.
"""
import ctypes as C

import torch

from . import capi
from .capi import MWError


def endrun(msg):
    """model/main_header.h:66-68"""
    raise MWError(msg)


class DataManager:
    """name -> tensor (+ dims, dirty flag, positive flag): DataManager.h:20-585."""

    def __init__(self, device):
        self.device = device
        self.entries = {}
        self.dimensions = {}
        # callables run in front of every access to an entry (get, the validators, clone_into): a module that keeps part of a field's value parked
        # elsewhere -- ColumnNudger's deferred increments, modules.py -- registers its flush here, so that whoever LOOKS at a field sees all of it
        self.before_access = []

    def _sync_entries(self):
        for fn in self.before_access:
            fn()

    def add_dimension(self, name, length):                      # DataManager.h:106-120
        if name in self.dimensions and self.dimensions[name] != length:
            endrun("ERROR: Attempting to add a dimension of the same name as an existing dimension but not the same size")
        self.dimensions[name] = int(length)

    def find_dimension(self, name):
        return 0 if name in self.dimensions else -1

    def get_dimension_size(self, name):
        if name not in self.dimensions:
            endrun("ERROR: Could not find dimension.")
        return self.dimensions[name]

    def register_and_allocate(self, name, desc, dims, dim_names=None, dtype=torch.float64, positive=False):   # :122-181
        if name == "":
            endrun("ERROR: You cannot register_and_allocate with an empty string")
        if name in self.entries:
            endrun("ERROR: Duplicate entry name: " + name)
        t = torch.zeros(tuple(int(d) for d in dims), dtype=dtype, device=self.device)
        self.entries[name] = dict(desc=desc, data=t, dim_names=list(dim_names or []), dirty=False, positive=positive)
        return t

    def entry_exists(self, name):
        return name in self.entries

    def unregister_and_deallocate(self, name):                  # :199-203
        self._find_entry_or_error(name)
        del self.entries[name]

    # ---- dirty flags (:206-237): set by every non-const get ----
    def clean_all_entries(self):
        for e in self.entries.values():
            e["dirty"] = False

    def clean_entry(self, name):
        self._find_entry_or_error(name)["dirty"] = False

    def entry_is_dirty(self, name):
        return self._find_entry_or_error(name)["dirty"]

    def get_dirty_entries(self):
        return [k for k, e in self.entries.items() if e["dirty"]]

    def _find_entry_or_error(self, name):                       # :505-511
        if name not in self.entries:
            endrun("ERROR: Attempting to retrieve variable name [" + name + "], but it doesn't exist. ")
        return self.entries[name]

    # ---- validators (:385-483).  The reference copies every array to the host ("This is EXPENSIVE"); here one device pass per entry
    # (mw_validate_f64 / _f32: counts and the first flat index of NaN / inf / negative elements).  A failed check prints the
    # reference's warning (first offending index) to stderr and ends the run when die_on_failed_check.  Returns the findings. ----
    def _scan(self, name):
        self._sync_entries()
        e = self._find_entry_or_error(name)
        t = e["data"]
        if t.dtype not in (torch.float64, torch.float32):       # integer / bool entries: no NaN / inf; negatives only for signed ints
            neg = (t < 0) if t.dtype in (torch.int8, torch.int16, torch.int32, torch.int64) else None
            n_neg = int(neg.sum()) if neg is not None else 0
            first = int(neg.view(-1).nonzero()[0]) if n_neg else -1
            return e, [0, 0, n_neg, -1, -1, first]
        out = (C.c_longlong * 6)()
        fn = capi.lib().mw_validate_f64 if t.dtype == torch.float64 else capi.lib().mw_validate_f32
        t = t.contiguous()                                      # (the scan walks numel() elements from data_ptr(): flat index = the reference's global index)
        with torch.cuda.device(t.device):                       # the scan allocates and launches on the CURRENT device: make it the entry's
            capi.check(fn(t.data_ptr(), t.numel(), out, torch.cuda.current_stream(t.device).cuda_stream))
        return e, list(out)

    def _report(self, what, name, count, first, die):
        if count:
            import sys
            print("WARNING: %s discovered in: %s at global index: %d  (%d element%s)" % (what, name, first, count, "" if count == 1 else "s"),
                  file=sys.stderr)
            if die:
                endrun("")
        return count

    def validate_nan(self, name, die_on_failed_check=False):    # :402-418
        _, r = self._scan(name)
        return self._report("NaN", name, r[0], r[3], die_on_failed_check)

    def validate_inf(self, name, die_on_failed_check=False):    # :421-428
        _, r = self._scan(name)
        return self._report("inf", name, r[1], r[4], die_on_failed_check)

    def validate_pos(self, name, die_on_failed_check=False):    # :431-441, :471-483: only entries registered positive
        e, r = self._scan(name)
        if not e["positive"]:
            return 0
        return self._report("negative value in positive-definite entry", name, r[2], r[5], die_on_failed_check)

    def validate(self, name, die_on_failed_check=False):        # :393-397 (one scan for the three checks)
        e, r = self._scan(name)
        bad = self._report("NaN", name, r[0], r[3], die_on_failed_check)
        bad += self._report("inf", name, r[1], r[4], die_on_failed_check)
        if e["positive"]:
            bad += self._report("negative value in positive-definite entry", name, r[2], r[5], die_on_failed_check)
        return bad

    def validate_all(self, die_on_failed_check=False):          # :385-387
        return sum(self.validate(n, die_on_failed_check) for n in list(self.entries))

    def get(self, name, readonly=False):                        # :246-286
        if name not in self.entries:
            endrun("ERROR: Could not find entry name: " + name)
        self._sync_entries()
        if not readonly:
            self.entries[name]["dirty"] = True
        return self.entries[name]["data"]

    def get_lev_col(self, name, readonly=False):                # :289-330 -> (nlev, ncol) view
        t = self.get(name, readonly)
        return t.view(t.shape[0], -1)

    def get_collapsed(self, name, readonly=False):              # :333-365
        return self.get(name, readonly).view(-1)

    def clone_into(self, other):                                # :79-103
        self._sync_entries()
        other.dimensions = dict(self.dimensions)
        other.entries = {k: dict(v, data=v["data"].clone()) for k, v in self.entries.items()}


class Coupler:
    """core::Coupler (coupler.h:17-493)."""

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        self.options = {}
        self.tracers = []                     # [dict(name, desc, positive, adds_mass)]
        self.dm = DataManager(self.device)
        self.xlen = self.ylen = self.zlen = -1.0
        self.dt_gcm = -1.0
        self.nranks, self.myrank = 1, 0
        self.grid = capi.Grid()

    # ---- decomposition (coupler.h:110-214) ----
    def distribute_mpi_and_allocate_coupled_state(self, nz, ny_glob, nx_glob, nens, nranks=1, myrank=0):
        g = self.grid
        capi.check(capi.lib().mw_decompose(int(nranks), int(myrank), int(nx_glob), int(ny_glob), C.byref(g)))
        g.nz, g.nens = int(nz), int(nens)
        self.nranks, self.myrank = int(nranks), int(myrank)
        self.dm.add_dimension("nens", nens)
        self.dm.add_dimension("x", g.nx)
        self.dm.add_dimension("y", g.ny)
        self.dm.add_dimension("z", nz)

    # ---- options (coupler.h:281-320, Options.h) ----
    def add_option(self, key, value):
        if key not in self.options:
            self.options[key] = value

    def set_option(self, key, value):
        self.options[key] = value

    def get_option(self, key, default=None):
        if key in self.options:
            return self.options[key]
        if default is None:
            endrun("ERROR: option not found: " + key)
        return default

    def option_exists(self, key):
        return key in self.options

    def delete_option(self, key):
        self.options.pop(key, None)

    # ---- grid getters (coupler.h:219-278) ----
    def set_grid(self, xlen, ylen, zlen):
        self.xlen, self.ylen, self.zlen = float(xlen), float(ylen), float(zlen)
        self.grid.xlen, self.grid.ylen, self.grid.zlen = self.xlen, self.ylen, self.zlen

    def get_xlen(self): return self.xlen
    def get_ylen(self): return self.ylen
    def get_zlen(self): return self.zlen
    def get_nranks(self): return self.nranks
    def get_myrank(self): return self.myrank
    def get_nens(self): return self.grid.nens
    def get_nx_glob(self): return self.grid.nx_glob
    def get_ny_glob(self): return self.grid.ny_glob
    def get_nproc_x(self): return self.grid.nproc_x
    def get_nproc_y(self): return self.grid.nproc_y
    def get_px(self): return self.grid.px
    def get_py(self): return self.grid.py
    def get_i_beg(self): return self.grid.i_beg
    def get_j_beg(self): return self.grid.j_beg
    def is_sim2d(self): return self.grid.ny_glob == 1
    def is_mainproc(self): return self.myrank == 0
    def get_neighbor_rankid_matrix(self): return [list(self.grid.neigh[r * 3:(r + 1) * 3]) for r in range(3)]
    def get_data_manager_readonly(self): return self.dm
    def get_data_manager_readwrite(self): return self.dm
    def get_nx(self): return self.dm.get_dimension_size("x") if self.dm.find_dimension("x") != -1 else -1
    def get_ny(self): return self.dm.get_dimension_size("y") if self.dm.find_dimension("y") != -1 else -1
    def get_nz(self): return self.dm.get_dimension_size("z") if self.dm.find_dimension("z") != -1 else -1
    def get_dx(self): return self.xlen / self.grid.nx_glob
    def get_dy(self): return self.ylen / self.grid.ny_glob
    def get_dz(self): return self.zlen / self.get_nz()
    def get_num_tracers(self): return len(self.tracers)

    # ---- tracers (coupler.h:323-362) ----
    def add_tracer(self, tracer_name, tracer_desc, positive, adds_mass):
        nz, ny, nx, nens = self.get_nz(), self.get_ny(), self.get_nx(), self.get_nens()
        self.dm.register_and_allocate(tracer_name, tracer_desc, (nz, ny, nx, nens), ["z", "y", "x", "nens"], positive=positive)
        self.tracers.append(dict(name=tracer_name, desc=tracer_desc, positive=bool(positive), adds_mass=bool(adds_mass)))

    def get_tracer_names(self):
        return [t["name"] for t in self.tracers]

    def get_tracer_info(self, tracer_name):
        for t in self.tracers:
            if t["name"] == tracer_name:
                return t["desc"], True, t["positive"], t["adds_mass"]
        return "", False, False, False

    def tracer_exists(self, tracer_name):
        return any(t["name"] == tracer_name for t in self.tracers)

    def clone_into(self, other):                                 # coupler.h:85-107
        import copy
        other.options = copy.deepcopy(self.options)
        other.tracers = copy.deepcopy(self.tracers)
        other.xlen, other.ylen, other.zlen, other.dt_gcm = self.xlen, self.ylen, self.zlen, self.dt_gcm
        other.nranks, other.myrank = self.nranks, self.myrank
        C.memmove(C.byref(other.grid), C.byref(self.grid), C.sizeof(capi.Grid))
        self.dm.clone_into(other.dm)
