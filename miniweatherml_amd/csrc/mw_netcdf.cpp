// =====================================================================================================
// mw_netcdf.cpp -- a minimal writer for the netCDF *classic* on-disk formats CDF-2 (64-bit offset) and CDF-5 (64-bit data),
// host only, no library dependency.  SURVEY.md 8(f) rank 2: the reference writes its output through PnetCDF with
// NC_CLOBBER | NC_64BIT_DATA, i.e. CDF-5 (dynamics_euler_stratified_wenofv.h:2106-2112, time_averager.h:104), dims
// x,y,z (+ unlimited t), coordinate variables and one double variable per field; every rank writes its (z, y-block, x-block)
// hyperslab of member 0 into the one shared file (write1_all, :2184).  Here: the same file layout, written with pwrite();
// any number of processes on a node may write disjoint hyperslabs after the creating rank has finished mw_nc_enddef().
//
// Format (netCDF classic format specification, "CDF-5" variant in brackets), everything big-endian:
//   header   = magic numrecs dim_list gatt_list var_list
//   magic    = 'C' 'D' 'F' version(2|5)         numrecs = INT32 [INT64]
//   dim_list = NC_DIMENSION(=10, INT32) nelems dim*   |  ABSENT = ZERO(INT32) ZERO(nelems)
//   dim      = name length                       name = nelems chars padded to a multiple of 4;  length 0 = record dimension
//   var_list = NC_VARIABLE(=11) nelems var*      var = name ndims dimid* vatt_list nc_type(INT32, 6 = double) vsize begin(INT64)
//   nelems, ndims, dimid, length, vsize are INT32 [INT64 in CDF-5]
//   data     = fixed-size variables at their `begin`, then the records: record r holds, for every record variable in
//              definition order, that variable's slab r (vsize bytes) at begin + r * recsize.
// Only int / float / double variables and no attributes are written -- which is all the reference writes.
// =====================================================================================================
#include "../../include/mw_cdna4.h"
#include "mw_common.h"
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

using mw::set_error;

namespace {

struct Dim { std::string name; long long len; };
struct Var { std::string name; std::vector<int> dims; long long vsize = 0, begin = 0; bool rec = false; int type = 6;
             int esize() const { return type == 6 ? 8 : 4; } };       // nc_type: 4 = NC_INT, 5 = NC_FLOAT, 6 = NC_DOUBLE

uint64_t be64(uint64_t v) { return __builtin_bswap64(v); }
uint32_t be32(uint32_t v) { return __builtin_bswap32(v); }

} // namespace

struct mw_nc_s {
  int fd = -1, format = 5;
  bool defining = true;
  long long header_align = 512, var_align = 4;
  long long numrecs = 0, recsize = 0, rec_begin = 0;
  std::vector<Dim> dims;
  std::vector<Var> vars;
  std::vector<unsigned char> swapbuf;
  void *stage_dev = nullptr, *stage_host = nullptr; size_t stage_cap = 0;   // mw_output_put_field's staging pair (grow-only, freed in close)

  void put32(std::vector<unsigned char> &b, uint32_t v) { uint32_t x = be32(v); b.insert(b.end(), (unsigned char *)&x, (unsigned char *)&x + 4); }
  void put64(std::vector<unsigned char> &b, uint64_t v) { uint64_t x = be64(v); b.insert(b.end(), (unsigned char *)&x, (unsigned char *)&x + 8); }
  void putn(std::vector<unsigned char> &b, long long v) { if (format == 5) put64(b, (uint64_t)v); else put32(b, (uint32_t)v); }   // NON_NEG
  void putname(std::vector<unsigned char> &b, const std::string &s) {
    putn(b, (long long)s.size());
    b.insert(b.end(), s.begin(), s.end());
    while (b.size() % 4) b.push_back(0);
  }
  std::vector<unsigned char> header() {
    std::vector<unsigned char> b = {'C', 'D', 'F', (unsigned char)format};
    putn(b, numrecs);
    if (dims.empty()) { put32(b, 0); putn(b, 0); }
    else { put32(b, 10); putn(b, (long long)dims.size()); for (auto &d : dims) { putname(b, d.name); putn(b, d.len); } }
    put32(b, 0); putn(b, 0);                                       // no global attributes
    if (vars.empty()) { put32(b, 0); putn(b, 0); }
    else {
      put32(b, 11); putn(b, (long long)vars.size());
      for (auto &v : vars) {
        putname(b, v.name);
        putn(b, (long long)v.dims.size());
        for (int d : v.dims) putn(b, d);
        put32(b, 0); putn(b, 0);                                   // no variable attributes
        put32(b, (uint32_t)v.type);                                // nc_type
        putn(b, v.vsize);
        put64(b, (uint64_t)v.begin);
      }
    }
    return b;
  }
};

static long long round_up(long long v, long long a) { return a > 1 ? ((v + a - 1) / a) * a : v; }

static int pwrite_all(int fd, const void *buf, size_t n, long long off) {
  const char *p = (const char *)buf;
  while (n) {
    ssize_t w = pwrite(fd, p, n, (off_t)off);
    if (w < 0) { if (errno == EINTR) continue; return 1; }
    p += w; n -= (size_t)w; off += w;
  }
  return 0;
}
static int pread_all(int fd, void *buf, size_t n, long long off) {
  char *p = (char *)buf;
  while (n) {
    ssize_t r = pread(fd, p, n, (off_t)off);
    if (r < 0) { if (errno == EINTR) continue; return 1; }
    if (r == 0) return 1;
    p += r; n -= (size_t)r; off += r;
  }
  return 0;
}

namespace mw {
// One device + one pinned host staging buffer per file handle for mw_output_put_field (mw_output.hip): allocating and freeing
// a pair per field and record (hipMalloc / hipHostMalloc are device-wide synchronisation points) serialised output-heavy runs.
int nc_staging(mw_nc_t nc, size_t bytes, double **dev, double **host) {
  if (nc->stage_cap < bytes) {
    if (nc->stage_dev) (void)hipFree(nc->stage_dev);
    if (nc->stage_host) (void)hipHostFree(nc->stage_host);
    nc->stage_dev = nc->stage_host = nullptr; nc->stage_cap = 0;
    if (hipMalloc(&nc->stage_dev, bytes) != hipSuccess) { nc->stage_dev = nullptr; MW_FAIL("output staging: hipMalloc failed"); }
    if (hipHostMalloc(&nc->stage_host, bytes, hipHostMallocDefault) != hipSuccess) {
      (void)hipFree(nc->stage_dev); nc->stage_dev = nc->stage_host = nullptr; MW_FAIL("output staging: hipHostMalloc failed"); }
    nc->stage_cap = bytes;
  }
  *dev = (double *)nc->stage_dev; *host = (double *)nc->stage_host;
  return 0;
}
} // namespace mw

extern "C" {

int mw_nc_create(mw_nc_t *out, const char *path, int format, long long header_align, long long var_align) {
  if (!out || !path) MW_FAIL("nc_create: null argument");
  if (format != 2 && format != 5) MW_FAIL("nc_create: format must be 2 (64-bit offset) or 5 (64-bit data)");
  int fd = open(path, O_CREAT | O_TRUNC | O_RDWR, 0644);                    // NC_CLOBBER
  if (fd < 0) MW_FAIL("nc_create: cannot create the file");
  mw_nc_s *nc = new mw_nc_s();
  nc->fd = fd; nc->format = format;
  nc->header_align = header_align > 0 ? header_align : 512;
  nc->var_align = var_align > 0 ? var_align : 4;
  *out = nc;
  return 0;
}

int mw_nc_def_dim(mw_nc_t nc, const char *name, long long len, int *dimid) {
  if (!nc || !name || !dimid) MW_FAIL("nc_def_dim: null argument");
  if (!nc->defining) MW_FAIL("nc_def_dim: not in define mode");
  if (len < 0) MW_FAIL("nc_def_dim: negative length");
  if (len == 0) for (auto &d : nc->dims) if (d.len == 0) MW_FAIL("nc_def_dim: only one record dimension is allowed");
  nc->dims.push_back({name, len});
  *dimid = (int)nc->dims.size() - 1;
  return 0;
}

int mw_nc_def_var_typed(mw_nc_t nc, const char *name, int nc_type, int ndims, const int *dimids, int *varid) {
  if (!nc || !name || !varid || (ndims > 0 && !dimids)) MW_FAIL("nc_def_var: null argument");
  if (!nc->defining) MW_FAIL("nc_def_var: not in define mode");
  if (nc_type != 4 && nc_type != 5 && nc_type != 6) MW_FAIL("nc_def_var: nc_type must be 4 (int), 5 (float) or 6 (double)");
  Var v; v.name = name; v.type = nc_type;
  for (int i = 0; i < ndims; i++) {
    if (dimids[i] < 0 || dimids[i] >= (int)nc->dims.size()) MW_FAIL("nc_def_var: bad dimension id");
    if (nc->dims[dimids[i]].len == 0) { if (i != 0) MW_FAIL("nc_def_var: the record dimension must come first"); v.rec = true; }
    v.dims.push_back(dimids[i]);
  }
  nc->vars.push_back(v);
  *varid = (int)nc->vars.size() - 1;
  return 0;
}

int mw_nc_def_var(mw_nc_t nc, const char *name, int ndims, const int *dimids, int *varid) {
  return mw_nc_def_var_typed(nc, name, 6, ndims, dimids, varid);
}

int mw_nc_enddef(mw_nc_t nc) {
  if (!nc) MW_FAIL("nc_enddef: null handle");
  if (!nc->defining) MW_FAIL("nc_enddef: not in define mode");
  for (auto &v : nc->vars) {
    long long n = v.esize();
    for (int d : v.dims) if (nc->dims[d].len > 0) n *= nc->dims[d].len;
    v.vsize = round_up(n, 4);
    if (nc->format == 2 && v.vsize > 0xFFFFFFFFll) MW_FAIL("nc_enddef: variable too large for CDF-2; use format 5");
  }
  long long off = round_up((long long)nc->header().size(), nc->header_align);           // nc_header_align_size
  for (auto &v : nc->vars) if (!v.rec) { off = round_up(off, nc->var_align); v.begin = off; off += v.vsize; }   // nc_var_align_size
  off = round_up(off, nc->var_align);
  nc->rec_begin = off; nc->recsize = 0;
  for (auto &v : nc->vars) if (v.rec) { v.begin = off; off += v.vsize; nc->recsize += v.vsize; }
  nc->defining = false;
  std::vector<unsigned char> h = nc->header();
  if (pwrite_all(nc->fd, h.data(), h.size(), 0)) MW_FAIL("nc_enddef: header write failed");
  // make the fixed part exist (zero-filled holes), so that a reader never runs past the end of the file
  if (ftruncate(nc->fd, (off_t)nc->rec_begin) != 0) MW_FAIL("nc_enddef: ftruncate failed");
  if (fsync(nc->fd) != 0) MW_FAIL("nc_enddef: fsync failed");
  return 0;
}

// Re-opens a file written by this library (any rank, after the creator's enddef): parses the header.
int mw_nc_open(mw_nc_t *out, const char *path) {
  if (!out || !path) MW_FAIL("nc_open: null argument");
  int fd = open(path, O_RDWR);
  if (fd < 0) MW_FAIL("nc_open: cannot open the file");
  mw_nc_s *nc = new mw_nc_s();
  nc->fd = fd; nc->defining = false;
  auto fail = [&](const char *m) { close(fd); delete nc; set_error(m); return 1; };
  unsigned char magic[4];
  if (pread_all(fd, magic, 4, 0) || magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F' || (magic[3] != 2 && magic[3] != 5))
    return fail("nc_open: not a CDF-2/CDF-5 file");
  nc->format = magic[3];
  long long pos = 4;
  bool bad = false;
  auto get32 = [&]() { uint32_t v = 0; if (pread_all(fd, &v, 4, pos)) bad = true; pos += 4; return (long long)be32(v); };
  auto get64 = [&]() { uint64_t v = 0; if (pread_all(fd, &v, 8, pos)) bad = true; pos += 8; return (long long)be64(v); };
  auto getn = [&]() { return nc->format == 5 ? get64() : get32(); };
  auto getname = [&]() { long long n = getn(); std::string s((size_t)(bad || n < 0 || n > 4096 ? 0 : n), '\0');
                         if (!s.empty() && pread_all(fd, &s[0], s.size(), pos)) bad = true; pos += round_up(n, 4); return s; };
  nc->numrecs = getn();
  long long tag = get32(), n = getn();
  if (tag != 0 && tag != 10) return fail("nc_open: bad dimension list");
  for (long long i = 0; i < n && !bad; i++) { Dim d; d.name = getname(); d.len = getn(); nc->dims.push_back(d); }
  tag = get32(); n = getn();
  if (tag != 0 || n != 0) return fail("nc_open: files with attributes are not supported");
  tag = get32(); n = getn();
  if (tag != 0 && tag != 11) return fail("nc_open: bad variable list");
  for (long long i = 0; i < n && !bad; i++) {
    Var v; v.name = getname();
    long long nd = getn();
    for (long long k = 0; k < nd && !bad; k++) { int d = (int)getn(); if (d < 0 || d >= (int)nc->dims.size()) bad = true; else v.dims.push_back(d); }
    long long atag = get32(), an = getn();
    if (atag != 0 || an != 0) return fail("nc_open: files with attributes are not supported");
    v.type = (int)get32();
    if (v.type != 4 && v.type != 5 && v.type != 6) return fail("nc_open: only int, float and double variables are supported");
    v.vsize = getn(); v.begin = get64();
    v.rec = !v.dims.empty() && nc->dims[v.dims[0]].len == 0;
    nc->vars.push_back(v);
  }
  if (bad) return fail("nc_open: truncated or corrupt header");
  nc->recsize = 0; nc->rec_begin = 0;
  for (auto &v : nc->vars) if (v.rec) { if (!nc->recsize) nc->rec_begin = v.begin; nc->recsize += v.vsize; }
  *out = nc;
  return 0;
}

int mw_nc_inq_varid(mw_nc_t nc, const char *name, int *varid) {
  if (!nc || !name || !varid) MW_FAIL("nc_inq_varid: null argument");
  for (size_t i = 0; i < nc->vars.size(); i++) if (nc->vars[i].name == name) { *varid = (int)i; return 0; }
  MW_FAIL("nc_inq_varid: no such variable");
}

int mw_nc_inq_dimlen(mw_nc_t nc, const char *name, long long *len) {
  if (!nc || !name || !len) MW_FAIL("nc_inq_dimlen: null argument");
  for (auto &d : nc->dims) if (d.name == name) {
    if (d.len == 0 && !nc->defining) {                               // current number of records, re-read from the file
      unsigned char b[8] = {0};
      const int w = nc->format == 5 ? 8 : 4;
      if (pread_all(nc->fd, b, w, 4)) MW_FAIL("nc_inq_dimlen: header read failed");
      uint64_t v = 0; for (int i = 0; i < w; i++) v = (v << 8) | b[i];
      nc->numrecs = (long long)v;
      *len = nc->numrecs;
    } else *len = d.len;
    return 0;
  }
  MW_FAIL("nc_inq_dimlen: no such dimension");
}

// Writes the hyperslab start[], count[] (one entry per dimension of the variable) from HOST data of the variable's own type
// (C order); values are byte-swapped to big-endian on the way.
int mw_nc_put_vara(mw_nc_t nc, int varid, const long long *start, const long long *count, const void *data) {
  if (!nc || !data) MW_FAIL("nc_put_vara: null argument");
  if (nc->defining) MW_FAIL("nc_put_vara: still in define mode");
  if (varid < 0 || varid >= (int)nc->vars.size()) MW_FAIL("nc_put_vara: bad variable id");
  const Var &v = nc->vars[varid];
  const int nd = (int)v.dims.size(), es = v.esize();
  if (nd > 0 && (!start || !count)) MW_FAIL("nc_put_vara: null argument");
  std::vector<long long> len(nd), stride(nd);
  for (int i = 0; i < nd; i++) {
    len[i] = nc->dims[v.dims[i]].len;
    if (start[i] < 0 || count[i] < 0 || (len[i] > 0 && start[i] + count[i] > len[i])) MW_FAIL("nc_put_vara: hyperslab out of range");
  }
  long long total = 1;
  for (int i = 0; i < nd; i++) total *= count[i];
  if (total == 0) return 0;
  // element strides inside one record (or inside the whole fixed variable)
  long long acc = 1;
  for (int i = nd - 1; i >= 0; i--) { stride[i] = acc; if (!(v.rec && i == 0)) acc *= len[i]; }
  const long long row = nd ? count[nd - 1] : 1;                      // contiguous run in the file
  nc->swapbuf.resize((size_t)row * es);
  std::vector<long long> idx(nd, 0);
  const unsigned char *src = (const unsigned char *)data;
  for (long long done = 0; done < total; done += row) {
    long long off = v.begin;
    for (int i = 0; i < nd; i++) {
      const long long c = start[i] + idx[i];
      off += (v.rec && i == 0) ? c * nc->recsize : c * stride[i] * es;
    }
    if (es == 8) { uint64_t *sb = (uint64_t *)nc->swapbuf.data();
                   for (long long r = 0; r < row; r++) { uint64_t u; memcpy(&u, src + r * 8, 8); sb[r] = be64(u); } }
    else         { uint32_t *sb = (uint32_t *)nc->swapbuf.data();
                   for (long long r = 0; r < row; r++) { uint32_t u; memcpy(&u, src + r * 4, 4); sb[r] = be32(u); } }
    if (pwrite_all(nc->fd, nc->swapbuf.data(), (size_t)row * es, off)) MW_FAIL("nc_put_vara: write failed");
    src += row * es;
    for (int i = nd - 2; i >= 0; i--) { if (++idx[i] < count[i]) break; idx[i] = 0; }
  }
  return 0;
}

int mw_nc_put_vara_double(mw_nc_t nc, int varid, const long long *start, const long long *count, const double *data) {
  if (nc && varid >= 0 && varid < (int)nc->vars.size() && nc->vars[varid].type != 6) MW_FAIL("nc_put_vara_double: not a double variable");
  return mw_nc_put_vara(nc, varid, start, count, data);
}

// Sets the record count in the header (the creating / main rank calls this after a record has been written).
int mw_nc_set_numrecs(mw_nc_t nc, long long numrecs) {
  if (!nc) MW_FAIL("nc_set_numrecs: null handle");
  if (nc->defining) MW_FAIL("nc_set_numrecs: still in define mode");
  nc->numrecs = numrecs;
  unsigned char b[8];
  const int w = nc->format == 5 ? 8 : 4;
  for (int i = 0; i < w; i++) b[i] = (unsigned char)((unsigned long long)numrecs >> (8 * (w - 1 - i)));
  if (pwrite_all(nc->fd, b, w, 4)) MW_FAIL("nc_set_numrecs: write failed");
  const long long end = nc->rec_begin + numrecs * nc->recsize;       // the file must cover every record completely
  struct { off_t sz; } cur; cur.sz = lseek(nc->fd, 0, SEEK_END);
  if (cur.sz < (off_t)end && ftruncate(nc->fd, (off_t)end) != 0) MW_FAIL("nc_set_numrecs: ftruncate failed");
  return 0;
}

int mw_nc_close(mw_nc_t nc) {
  if (!nc) return 0;
  int rc = 0;
  if (nc->stage_dev) (void)hipFree(nc->stage_dev);
  if (nc->stage_host) (void)hipHostFree(nc->stage_host);
  if (nc->fd >= 0) { if (fsync(nc->fd) != 0) rc = 1; if (close(nc->fd) != 0) rc = 1; }
  delete nc;
  if (rc) MW_FAIL("nc_close: fsync/close failed");
  return 0;
}

} // extern "C"
