// =====================================================================================================
// mw_h5.cpp -- a minimal, dependency-free reader for 32-bit float datasets of an HDF5 file, host only.
// Replaces ponni::load_h5_weights<N>(file, group, dataset) at its call sites in the surrogate module
// (experiments/supercell_kessler_surrogate/custom_modules/microphysics_kessler_ponni.h:103-107), which reads the Keras weight
// file `keras_weights_h5` (Dense kernels (in,out) and biases, H5T_IEEE_F32LE).  There is no HDF5 library on the target boxes.
//
// Supported subset (what h5py / Keras 2.x write with default settings, and what the shipped file uses):
//   superblock version 0 or 1, 8-byte offsets and lengths; "old style" groups (symbol-table message: v1 B-tree of
//   symbol-table nodes + local heap); version-1 object headers with continuation blocks; dataspace message v1 / v2 (simple);
//   datatype class 1 (IEEE float), 4 bytes, little-endian; data layout message v3 contiguous or compact (and v1 / v2 contiguous).
// Anything else (chunked / filtered data, new-style groups, other datatypes) is refused with a message -- never guessed.
// Format: HDF5 File Format Specification, version 2.0 (sections II.A superblock, III.A B-trees, III.B symbol table nodes,
// III.C symbol table entries, III.D local heaps, IV.A object headers and messages 0x0001, 0x0003, 0x0008, 0x0010, 0x0011).
// =====================================================================================================
#include "../../include/mw_cdna4.h"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace mw { void set_error(const std::string &msg); }

namespace {

struct H5File {
  std::vector<unsigned char> b;
  std::string err;
  bool fail(const std::string &m) { if (err.empty()) err = m; return false; }
  bool in(uint64_t off, uint64_t n) const { return off <= b.size() && n <= b.size() - off; }
  uint64_t u(uint64_t off, int n) {                          // little-endian unsigned of n bytes
    if (!in(off, (uint64_t)n)) { fail("h5: read beyond the end of the file"); return 0; }
    uint64_t v = 0;
    for (int i = n - 1; i >= 0; i--) v = (v << 8) | b[off + i];
    return v;
  }
  int byte(uint64_t off) { return (int)u(off, 1); }          // one byte, bounds-checked like everything that comes from the file
  bool sig(uint64_t off, const char *s4) { return in(off, 4) && memcmp(&b[off], s4, 4) == 0; }
};

struct ObjInfo {                                             // what the messages of one object header say
  bool has_symtab = false; uint64_t btree = 0, heap = 0;
  bool has_space = false; std::vector<long long> dims;
  bool has_type = false, type_ok = false; std::string type_desc;
  bool has_layout = false; int layout_class = -1; uint64_t data_addr = 0, data_size = 0;     // compact: data_addr = file offset of the bytes
};

bool parse_messages(H5File &f, uint64_t off, uint64_t len, int &remaining, ObjInfo &o, int depth);

bool parse_header(H5File &f, uint64_t addr, ObjInfo &o) {
  if (!f.in(addr, 16)) return f.fail("h5: object header address beyond the end of the file");
  if (f.byte(addr) != 1) return f.fail("h5: only version-1 object headers are supported (this file uses a newer object header)");
  int nmsg = (int)f.u(addr + 2, 2);
  uint64_t hsize = f.u(addr + 8, 4);
  return parse_messages(f, addr + 16, hsize, nmsg, o, 0);
}

bool parse_messages(H5File &f, uint64_t off, uint64_t len, int &remaining, ObjInfo &o, int depth) {
  if (depth > 16) return f.fail("h5: object header continuation chain too deep");
  const uint64_t end = off + len;
  while (remaining > 0 && off + 8 <= end) {
    const int type = (int)f.u(off, 2);
    const uint64_t size = f.u(off + 2, 2);
    const uint64_t d = off + 8;
    if (!f.in(d, size)) return f.fail("h5: header message beyond the end of the file");
    remaining--;
    if (type == 0x0011) {                                    // symbol table message: this object is an old-style group
      if (size < 16) return f.fail("h5: symbol table message too short");
      o.has_symtab = true; o.btree = f.u(d, 8); o.heap = f.u(d + 8, 8);
    } else if (type == 0x0001) {                             // dataspace
      if (size < 4) return f.fail("h5: dataspace message too short");
      const int ver = f.byte(d), rank = f.byte(d + 1);
      uint64_t p = (ver == 1) ? d + 8 : d + 4;
      if (ver != 1 && ver != 2) return f.fail("h5: unsupported dataspace message version");
      if (ver == 2 && f.byte(d + 3) == 2) return f.fail("h5: null dataspace");
      if (rank > 8 || !f.in(p, 8 * (uint64_t)rank)) return f.fail("h5: dataspace rank / extent beyond the message");
      o.dims.clear();
      for (int r = 0; r < rank; r++) o.dims.push_back((long long)f.u(p + 8 * (uint64_t)r, 8));
      o.has_space = true;
    } else if (type == 0x0003) {                             // datatype
      if (size < 8) return f.fail("h5: datatype message too short");
      const int cls = f.byte(d) & 0x0F;
      const int bits0 = f.byte(d + 1);
      const uint64_t tsize = f.u(d + 4, 4);
      o.has_type = true;
      o.type_ok = (cls == 1 && tsize == 4 && (bits0 & 1) == 0);       // floating point, 4 bytes, little-endian
      o.type_desc = "class " + std::to_string(cls) + ", " + std::to_string(tsize) + " bytes" + ((bits0 & 1) ? ", big-endian" : "");
    } else if (type == 0x0008) {                             // data layout
      if (size < 4) return f.fail("h5: data layout message too short");
      const int ver = f.byte(d);
      o.has_layout = true;
      if (ver == 3) {
        o.layout_class = f.byte(d + 1);
        if (o.layout_class == 0) { o.data_size = f.u(d + 2, 2); o.data_addr = d + 4; }
        else if (o.layout_class == 1) { o.data_addr = f.u(d + 2, 8); o.data_size = f.u(d + 10, 8); }
      } else if (ver == 1 || ver == 2) {
        const int rank = f.byte(d + 1);
        o.layout_class = f.byte(d + 2);
        if (rank > 9) return f.fail("h5: data layout rank out of range");
        if (o.layout_class == 1) {                           // contiguous: address, then `rank` 4-byte dimension sizes (bytes = their product)
          o.data_addr = f.u(d + 8, 8);
          uint64_t n = 1;
          for (int r = 0; r < rank; r++) {
            const uint64_t e = f.u(d + 16 + 4 * (uint64_t)r, 4);
            if (e != 0 && n > (uint64_t)f.b.size() / e) return f.fail("h5: data layout larger than the file");
            n *= e;
          }
          o.data_size = n;
        } else if (o.layout_class == 0) {                    // compact: dimension sizes, then the size of the data and the data
          const uint64_t q = d + 8 + 4 * (uint64_t)rank;
          o.data_size = f.u(q, 4); o.data_addr = q + 4;
        }
      } else return f.fail("h5: unsupported data layout message version " + std::to_string(ver));
    } else if (type == 0x0010) {                             // continuation: more messages elsewhere
      if (size < 16) return f.fail("h5: continuation message too short");
      const uint64_t coff = f.u(d, 8), clen = f.u(d + 8, 8);
      if (!f.in(coff, clen)) return f.fail("h5: continuation block beyond the end of the file");
      if (!parse_messages(f, coff, clen, remaining, o, depth + 1)) return false;
    }
    off = d + size;
  }
  return f.err.empty();
}

// child `name` of the old-style group whose symbol table is (btree, heap): walks the v1 B-tree (node type 0) down to the symbol
// table nodes and compares the link names in the local heap
bool find_child(H5File &f, uint64_t btree, uint64_t heap, const std::string &name, uint64_t &obj_addr) {
  if (!f.sig(heap, "HEAP")) return f.fail("h5: local heap signature not found");
  const uint64_t heap_data = f.u(heap + 24, 8), heap_size = f.u(heap + 8, 8);
  if (!f.err.empty() || !f.in(heap_data, heap_size)) return f.fail("h5: local heap data segment beyond the end of the file");
  std::vector<uint64_t> stack{btree};
  int guard = 0;
  while (!stack.empty()) {
    if (++guard > 100000) return f.fail("h5: group B-tree does not terminate");
    const uint64_t node = stack.back(); stack.pop_back();
    if (f.sig(node, "TREE")) {
      if (f.byte(node + 4) != 0) return f.fail("h5: unexpected B-tree node type in a group");
      const int used = (int)f.u(node + 6, 2);
      uint64_t p = node + 24;                                // key 0
      for (int e = 0; e < used; e++) { stack.push_back(f.u(p + 8, 8)); p += 16; }      // (key, child) pairs: child after each key
    } else if (f.sig(node, "SNOD")) {
      const int nsym = (int)f.u(node + 6, 2);
      for (int e = 0; e < nsym; e++) {
        const uint64_t ent = node + 8 + 40 * (uint64_t)e;
        const uint64_t noff = f.u(ent, 8);
        if (noff >= heap_size) return f.fail("h5: link name offset outside the local heap");
        const char *s = (const char *)&f.b[heap_data + noff];
        const size_t maxlen = (size_t)(heap_size - noff);
        if (strnlen(s, maxlen) == name.size() && memcmp(s, name.data(), name.size()) == 0) { obj_addr = f.u(ent + 8, 8); return f.err.empty(); }
      }
    } else return f.fail("h5: neither a B-tree node nor a symbol table node where one was expected");
    if (!f.err.empty()) return false;
  }
  return f.fail("h5: no object named '" + name + "'");
}

} // namespace

extern "C" int mw_h5_read_f32(const char *file, const char *group, const char *dataset, float *out, long long capacity,
                              long long *dims, int *ndims) {
  auto bad = [](const std::string &m) { mw::set_error(m); return 1; };
  if (!file || !group || !dataset) return bad("h5_read_f32: null argument");
  H5File f;
  {
    FILE *fp = fopen(file, "rb");
    if (!fp) return bad(std::string("h5_read_f32: cannot open ") + file);
    fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
    if (n < 0 || n > (1L << 30)) { fclose(fp); return bad("h5_read_f32: unreasonable file size"); }
    f.b.resize((size_t)n);
    const size_t got = n ? fread(f.b.data(), 1, (size_t)n, fp) : 0;
    fclose(fp);
    if (got != (size_t)n) return bad(std::string("h5_read_f32: short read of ") + file);
  }
  static const unsigned char SIG[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
  if (f.b.size() < 96 || memcmp(f.b.data(), SIG, 8) != 0) return bad(std::string(file) + " is not an HDF5 file (no signature at offset 0)");
  const int sver = f.b[8];
  if (sver != 0 && sver != 1) return bad("h5_read_f32: only superblock versions 0 and 1 are supported");
  if (f.b[13] != 8 || f.b[14] != 8) return bad("h5_read_f32: only 8-byte offsets and lengths are supported");
  const uint64_t root_entry = (sver == 0) ? 56 : 60;        // v1 has 4 more bytes (indexed-storage K + reserved)
  const uint64_t base = f.u(24 + (sver == 1 ? 4 : 0), 8);
  if (base != 0) return bad("h5_read_f32: non-zero base address");
  uint64_t obj = f.u(root_entry + 8, 8);
  // walk the path group/dataset
  std::string path = std::string(group) + "/" + dataset;
  size_t pos = 0;
  ObjInfo o;
  while (pos <= path.size()) {
    size_t nxt = path.find('/', pos);
    if (nxt == std::string::npos) nxt = path.size();
    const std::string comp = path.substr(pos, nxt - pos);
    pos = nxt + 1;
    if (comp.empty()) continue;
    o = ObjInfo();
    if (!parse_header(f, obj, o)) return bad(f.err);
    if (!o.has_symtab) return bad("h5_read_f32: '" + comp + "' is looked up in an object that is not an old-style group");
    if (!find_child(f, o.btree, o.heap, comp, obj)) return bad(f.err + " (path " + path + ")");
  }
  o = ObjInfo();
  if (!parse_header(f, obj, o)) return bad(f.err);
  if (!o.has_space || !o.has_type || !o.has_layout) return bad("h5_read_f32: " + path + " is not a dataset");
  if (!o.type_ok) return bad("h5_read_f32: " + path + " is not little-endian 32-bit float (" + o.type_desc + ")");
  if (o.layout_class != 0 && o.layout_class != 1) return bad("h5_read_f32: " + path + " is stored chunked; only contiguous / compact data are supported");
  if ((int)o.dims.size() > 8) return bad("h5_read_f32: rank > 8");
  long long n = 1;
  for (long long d : o.dims) {                               // untrusted: positive, and the product no larger than the file could hold
    if (d <= 0) return bad("h5_read_f32: " + path + " has a non-positive dimension");
    if (n > (long long)(f.b.size() / 4) / d) return bad("h5_read_f32: " + path + ": shape larger than the file");
    n *= d;
  }
  if (ndims) *ndims = (int)o.dims.size();
  if (dims) for (size_t r = 0; r < o.dims.size(); r++) dims[r] = o.dims[r];
  if (!out) return 0;                                        // shape query only
  if (n > capacity) return bad("h5_read_f32: " + path + " holds " + std::to_string(n) + " values, the buffer " + std::to_string(capacity));
  if (o.data_size < (uint64_t)n * 4 || !f.in(o.data_addr, (uint64_t)n * 4)) return bad("h5_read_f32: " + path + ": data extent beyond the file / smaller than its shape");
  memcpy(out, &f.b[o.data_addr], (size_t)n * 4);             // (host is little-endian: x86-64)
  return 0;
}
