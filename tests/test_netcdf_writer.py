"""The classic-netCDF writer behind the file output (host only, runs without a GPU): files written through the C ABI are read
back with (a) the independent reader tests/cdf.py (CDF-2 and CDF-5) and (b) scipy.io.netcdf_file (CDF-2 only: scipy predates
CDF-5), which is a third-party implementation of the same specification."""
import ctypes as C
import os

import numpy as np
import pytest

import cdf


@pytest.fixture(scope="module")
def L():
    from miniweatherml_amd import capi
    return capi.lib()


def write_file(L, path, fmt, nx, ny, nz, nrec, blocks=((0, 0),), halign=0, valign=0, rng=None):
    """The reference's layout: dims x,y,z,t ; vars x,y,z,t, then (t,z,y,x) fields.  `blocks`: (j_beg, i_beg) of the hyperslabs
    written separately (as the ranks of a 2-D decomposition would), each by a SECOND handle opened on the file."""
    from miniweatherml_amd.capi import check
    rng = rng or np.random.default_rng(3)
    nc = C.c_void_p()
    check(L.mw_nc_create(C.byref(nc), path.encode(), fmt, halign, valign))
    ids = {}
    for n, ln in (("x", nx), ("y", ny), ("z", nz), ("t", 0)):
        d = C.c_int()
        check(L.mw_nc_def_dim(nc, n.encode(), ln, C.byref(d)))
        ids[n] = d.value
    names = ["x", "y", "z", "t", "density_dry", "uvel", "water_vapor"]
    for n in names:
        dims = [ids[n]] if n in ids else [ids["t"], ids["z"], ids["y"], ids["x"]]
        v = C.c_int()
        check(L.mw_nc_def_var(nc, n.encode(), len(dims), (C.c_int * len(dims))(*dims), C.byref(v)))
    check(L.mw_nc_enddef(nc))
    data = {n: rng.normal(size=(nrec, nz, ny, nx)) for n in names[4:]}
    coords = {"x": (np.arange(nx) + 0.5) * 10.0, "y": (np.arange(ny) + 0.5) * 20.0, "z": (np.arange(nz) + 0.5) * 30.0}

    def put(h, name, start, count, arr):
        v = C.c_int()
        check(L.mw_nc_inq_varid(h, name.encode(), C.byref(v)))
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        check(L.mw_nc_put_vara_double(h, v.value, (C.c_longlong * len(start))(*start), (C.c_longlong * len(count))(*count),
                                      arr.ctypes.data_as(C.c_void_p)))
    for n in "xyz":
        put(nc, n, [0], [len(coords[n])], coords[n])
    jb = sorted({b[0] for b in blocks}) + [ny]
    ib = sorted({b[1] for b in blocks}) + [nx]
    for r in range(nrec):
        put(nc, "t", [r], [1], [1.5 * r])
        for (j0, i0) in blocks:
            j1, i1 = jb[jb.index(j0) + 1], ib[ib.index(i0) + 1]
            h = C.c_void_p()
            check(L.mw_nc_open(C.byref(h), path.encode()))                   # another "rank"
            for n in names[4:]:
                put(h, n, [r, 0, j0, i0], [1, nz, j1 - j0, i1 - i0], data[n][r, :, j0:j1, i0:i1])
            check(L.mw_nc_close(h))
        check(L.mw_nc_set_numrecs(nc, r + 1))
        n_now = C.c_longlong()
        check(L.mw_nc_inq_dimlen(nc, b"t", C.byref(n_now)))
        assert n_now.value == r + 1
    check(L.mw_nc_close(nc))
    return data, coords


@pytest.mark.parametrize("fmt", [2, 5])
@pytest.mark.parametrize("blocks", [((0, 0),), ((0, 0), (0, 7), (3, 0), (3, 7))])
def test_roundtrip_with_the_independent_reader(L, tmp_path, fmt, blocks):
    path = str(tmp_path / "out.nc")
    data, coords = write_file(L, path, fmt, 13, 5, 4, 3, blocks)
    r = cdf.Reader(path)
    assert r.version == fmt and r.numrecs == 3
    assert r.dims == [("x", 13), ("y", 5), ("z", 4), ("t", 0)]
    assert [v["name"] for v in r.vars] == ["x", "y", "z", "t", "density_dry", "uvel", "water_vapor"]
    assert all(v["type"] == 6 and v["atts"] == [] for v in r.vars) and r.gatts == []
    for n in "xyz":
        assert np.array_equal(r.get(n), coords[n])
    assert np.array_equal(r.get("t"), [0.0, 1.5, 3.0])
    for n, a in data.items():
        assert np.array_equal(r.get(n), a), n                                 # bit-exact: doubles are only byte-swapped
    # layout rules of the format: vsize = bytes of one record slab; record variables follow each other inside a record
    rec = [v for v in r.vars if r.is_rec(v)]
    assert rec[0]["vsize"] == 8 and all(v["vsize"] == 8 * 4 * 5 * 13 for v in rec[1:])
    assert all(b["begin"] == a["begin"] + a["vsize"] for a, b in zip(rec[:-1], rec[1:]))
    assert os.path.getsize(path) == rec[0]["begin"] + 3 * r.recsize()


def test_cdf2_file_is_read_by_scipy(L, tmp_path):
    scipy_io = pytest.importorskip("scipy.io")
    path = str(tmp_path / "out2.nc")
    data, coords = write_file(L, path, 2, 9, 6, 5, 2)
    f = scipy_io.netcdf_file(path, "r", mmap=False)
    assert f.version_byte == 2 and f.dimensions == {"x": 9, "y": 6, "z": 5, "t": None}
    assert f.variables["density_dry"].dimensions == ("t", "z", "y", "x") and f.variables["t"].isrec
    for n in "xyz":
        assert np.array_equal(f.variables[n][:], coords[n])
    for n, a in data.items():
        assert np.array_equal(f.variables[n][:], a), n
    f.close()


def test_alignment_hints_of_the_reference(L, tmp_path):
    """nc_header_align_size = nc_var_align_size = 1 MiB (dynamics_euler_stratified_wenofv.h:2103-2104)."""
    path = str(tmp_path / "aligned.nc")
    data, _ = write_file(L, path, 5, 6, 3, 2, 1, halign=1 << 20, valign=1 << 20)
    r = cdf.Reader(path)
    fixed = [v for v in r.vars if not r.is_rec(v)]
    assert [v["begin"] for v in fixed] == [1 << 20, 2 << 20, 3 << 20]
    assert next(v for v in r.vars if r.is_rec(v))["begin"] == 4 << 20
    assert np.array_equal(r.get("uvel"), data["uvel"])


def test_errors(L, tmp_path):
    from miniweatherml_amd.capi import MWError, check
    nc = C.c_void_p()
    with pytest.raises(MWError):
        check(L.mw_nc_create(C.byref(nc), str(tmp_path / "x.nc").encode(), 4, 0, 0))         # NetCDF-4/HDF5 is not provided
    with pytest.raises(MWError):
        check(L.mw_nc_open(C.byref(nc), str(tmp_path / "missing.nc").encode()))
    (tmp_path / "junk.nc").write_bytes(b"HDF\x89 not classic")
    with pytest.raises(MWError):
        check(L.mw_nc_open(C.byref(nc), str(tmp_path / "junk.nc").encode()))
    check(L.mw_nc_create(C.byref(nc), str(tmp_path / "y.nc").encode(), 5, 0, 0))
    d, v = C.c_int(), C.c_int()
    check(L.mw_nc_def_dim(nc, b"t", 0, C.byref(d)))
    with pytest.raises(MWError):
        check(L.mw_nc_def_dim(nc, b"t2", 0, C.byref(d)))                                      # one record dimension only
    check(L.mw_nc_def_dim(nc, b"x", 4, C.byref(d)))
    with pytest.raises(MWError):
        check(L.mw_nc_def_var(nc, b"bad", 2, (C.c_int * 2)(1, 0), C.byref(v)))                # record dimension must be first
    check(L.mw_nc_def_var(nc, b"a", 1, (C.c_int * 1)(1), C.byref(v)))
    buf = np.zeros(8)
    with pytest.raises(MWError):
        check(L.mw_nc_put_vara_double(nc, v.value, (C.c_longlong * 1)(0), (C.c_longlong * 1)(4), buf.ctypes.data_as(C.c_void_p)))   # define mode
    check(L.mw_nc_enddef(nc))
    with pytest.raises(MWError):
        check(L.mw_nc_put_vara_double(nc, v.value, (C.c_longlong * 1)(2), (C.c_longlong * 1)(4), buf.ctypes.data_as(C.c_void_p)))   # out of range
    check(L.mw_nc_close(nc))


@pytest.mark.parametrize("fmt", [2, 5])
def test_typed_variables_and_scalars(L, tmp_path, fmt):
    """The surrogate sample files: float record variables, double scalars, one int scalar (generate_micro_surrogate_data.h)."""
    from miniweatherml_amd.capi import MWError, check
    path = str(tmp_path / "typed.nc")
    nc = C.c_void_p()
    check(L.mw_nc_create(C.byref(nc), path.encode(), fmt, 0, 0))
    ids = {}
    for n, ln in (("nsamples", 0), ("num_vars_in", 5), ("sten_size", 2), ("num_vars_out", 4)):
        d = C.c_int()
        check(L.mw_nc_def_dim(nc, n.encode(), ln, C.byref(d)))
        ids[n] = d.value
    v = C.c_int()
    vid = {}
    for name, ty, dims in (("dx", 6, []), ("only_two_dimensions", 4, []), ("inputs", 5, [ids["nsamples"], ids["num_vars_in"], ids["sten_size"]]),
                           ("outputs", 5, [ids["nsamples"], ids["num_vars_out"]])):
        check(L.mw_nc_def_var_typed(nc, name.encode(), ty, len(dims), (C.c_int * max(1, len(dims)))(*dims), C.byref(v)))
        vid[name] = v.value
    with pytest.raises(MWError):
        check(L.mw_nc_def_var_typed(nc, b"bad", 2, 0, (C.c_int * 1)(0), C.byref(v)))                 # NC_CHAR is not provided
    check(L.mw_nc_enddef(nc))
    rng = np.random.default_rng(8)
    ins, outs = rng.normal(size=(7, 5, 2)).astype(np.float32), rng.normal(size=(7, 4)).astype(np.float32)

    def put(name, start, count, arr):
        check(L.mw_nc_put_vara(nc, vid[name], (C.c_longlong * max(1, len(start)))(*start), (C.c_longlong * max(1, len(count)))(*count),
                               arr.ctypes.data_as(C.c_void_p)))
    put("dx", [], [], np.array([250.0]))
    put("only_two_dimensions", [], [], np.array([1], dtype=np.int32))
    put("inputs", [0, 0, 0], [4, 5, 2], ins[:4]); put("outputs", [0, 0], [4, 4], outs[:4])
    check(L.mw_nc_set_numrecs(nc, 4))
    put("inputs", [4, 0, 0], [3, 5, 2], ins[4:]); put("outputs", [4, 0], [3, 4], outs[4:])        # a later time step appends
    check(L.mw_nc_set_numrecs(nc, 7))
    with pytest.raises(MWError):
        check(L.mw_nc_put_vara_double(nc, vid["inputs"], (C.c_longlong * 3)(0, 0, 0), (C.c_longlong * 3)(1, 5, 2), np.zeros(10).ctypes.data_as(C.c_void_p)))
    check(L.mw_nc_close(nc))
    r = cdf.Reader(path)
    assert r.numrecs == 7 and [(v["name"], v["type"]) for v in r.vars] == [("dx", 6), ("only_two_dimensions", 4), ("inputs", 5), ("outputs", 5)]
    assert r.get("dx") == 250.0 and r.get("only_two_dimensions") == 1
    assert np.array_equal(r.get("inputs"), ins) and np.array_equal(r.get("outputs"), outs)
    if fmt == 2:
        scipy_io = pytest.importorskip("scipy.io")
        f = scipy_io.netcdf_file(path, "r", mmap=False)
        assert np.array_equal(f.variables["inputs"][:], ins) and np.array_equal(f.variables["outputs"][:], outs)
        assert f.variables["only_two_dimensions"].getValue() == 1 and f.variables["dx"].getValue() == 250.0
        f.close()
