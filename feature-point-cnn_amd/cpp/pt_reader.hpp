// pt_reader.hpp -- reads the tensors of a PyTorch checkpoint file without libtorch.
//
// The reference writes its snapshots with torch.save (python/src/saveutils.py:54-63):
// a ZIP container (entries STORED, not deflated) holding `<name>/data.pkl` -- a
// protocol-2 pickle of {'epoch', 'model_state_dict', 'optimizer_state_dict',
// 'scaler_state_dict'} -- and one raw little-endian blob `<name>/data/<key>` per tensor
// storage.  The flat {name: tensor} files the reference's C++ loader expects
// (cpp/src/superpoint.cc:27-55, written by python/src/inferencewrapper.py:89-91) use the
// same container.  This header parses exactly that much: the ZIP central directory
// (incl. ZIP64 records), the pickle opcodes torch emits, and the two reduce calls that
// rebuild tensors.  Nothing here executes pickled code.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace fpc_pt {

struct Tensor {
  std::string dtype;              // "float32", "int64", ...
  std::vector<int64_t> shape;
  const void* data = nullptr;     // points into Checkpoint::bytes
  size_t numel = 0;
};

struct Checkpoint {
  std::vector<char> bytes;                       // whole file
  std::vector<std::pair<std::string, Tensor>> tensors;  // state-dict order
  const Tensor* find(const std::string& k) const {
    for (auto& kv : tensors)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
};

namespace detail {

inline uint16_t rd16(const char* p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint32_t rd32(const char* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const char* p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct ZipEntry { uint64_t offset = 0, size = 0; };

// name -> (offset of the entry's data, size) for STORED entries
// All offsets and lengths below come from the file: every range check is of the form `len <= n && off <= n - len`
// (no sum of untrusted values that could wrap).
inline std::map<std::string, ZipEntry> zip_directory(const std::vector<char>& f) {
  const size_t n = f.size();
  auto in_range = [n](uint64_t off, uint64_t len) { return len <= n && off <= n - len; };
  if (n < 22) throw std::runtime_error("not a zip file (too short)");
  size_t eocd = std::string::npos;
  for (size_t i = n - 22;; --i) {
    if (rd32(&f[i]) == 0x06054b50u) { eocd = i; break; }
    if (i == 0 || n - i > 22 + 65536) break;
  }
  if (eocd == std::string::npos) throw std::runtime_error("not a zip file (no end-of-central-directory record)");
  uint64_t count = rd16(&f[eocd + 10]), cd_size = rd32(&f[eocd + 12]), cd_off = rd32(&f[eocd + 16]);
  if (count == 0xffff || cd_off == 0xffffffffu || cd_size == 0xffffffffu) {  // ZIP64
    if (eocd < 20 || rd32(&f[eocd - 20]) != 0x07064b50u) throw std::runtime_error("zip64 locator missing");
    const uint64_t z = rd64(&f[eocd - 20 + 8]);
    if (!in_range(z, 56) || rd32(&f[z]) != 0x06064b50u) throw std::runtime_error("zip64 record missing");
    count = rd64(&f[z + 32]);
    cd_size = rd64(&f[z + 40]);
    cd_off = rd64(&f[z + 48]);
  }
  std::map<std::string, ZipEntry> dir;
  uint64_t p = cd_off;
  for (uint64_t i = 0; i < count; ++i) {
    if (!in_range(p, 46) || rd32(&f[p]) != 0x02014b50u) throw std::runtime_error("bad central directory entry");
    const uint16_t method = rd16(&f[p + 10]);
    uint64_t csize = rd32(&f[p + 20]), usize = rd32(&f[p + 24]), lho = rd32(&f[p + 42]);
    const uint16_t nlen = rd16(&f[p + 28]), xlen = rd16(&f[p + 30]), clen = rd16(&f[p + 32]);
    // name, extra field and comment follow the 46 fixed bytes; p + 46 <= n was just checked and the three lengths are
    // 16-bit, so these sums cannot wrap
    if (!in_range(p + 46, (uint64_t)nlen + xlen + clen)) throw std::runtime_error("central directory entry runs past the file");
    std::string name(&f[p + 46], nlen);
    uint64_t x = p + 46 + nlen;
    const uint64_t xend = x + xlen;
    while (x + 4 <= xend) {  // zip64 extended information
      const uint16_t id = rd16(&f[x]), sz = rd16(&f[x + 2]);
      if (id == 0x0001) {
        uint64_t q = x + 4;
        const uint64_t qend = x + 4 + sz < xend ? x + 4 + sz : xend;
        auto take64 = [&](uint64_t* dst) {
          if (q + 8 > qend) throw std::runtime_error("zip64 extra field of '" + name + "' is too short");
          *dst = rd64(&f[q]);
          q += 8;
        };
        if (usize == 0xffffffffu) take64(&usize);
        if (csize == 0xffffffffu) take64(&csize);
        if (lho == 0xffffffffu) take64(&lho);
      }
      x += 4 + (uint64_t)sz;
    }
    if (method != 0) throw std::runtime_error("zip entry '" + name + "' is compressed; torch.save stores entries raw");
    if (!in_range(lho, 30) || rd32(&f[lho]) != 0x04034b50u) throw std::runtime_error("bad local header for " + name);
    const uint64_t data = lho + 30 + rd16(&f[lho + 26]) + rd16(&f[lho + 28]);   // lho + 30 <= n: no wrap
    if (!in_range(data, usize)) throw std::runtime_error("zip entry out of range: " + name);
    dir[name] = ZipEntry{data, usize};
    p = xend + clen;
  }
  return dir;
}

// ---- a value model just rich enough for checkpoints --------------------------------------
struct Val;
using VP = std::shared_ptr<Val>;
struct Val {
  enum K { NONE, BOOL, INT, FLOAT, STR, TUPLE, LIST, DICT, GLOBAL, STORAGE, TENSOR, OBJ } k = NONE;
  int64_t i = 0;
  double d = 0;
  std::string s;                       // STR / GLOBAL "module name" / STORAGE key
  std::string dtype;                   // STORAGE / TENSOR
  std::vector<VP> items;               // TUPLE / LIST
  std::vector<std::pair<VP, VP>> dict; // DICT
  std::vector<int64_t> shape, stride;  // TENSOR
  int64_t offset = 0;
};
inline VP mk(Val::K k) { auto v = std::make_shared<Val>(); v->k = k; return v; }

inline std::string storage_dtype(const std::string& g) {
  static const std::pair<const char*, const char*> t[] = {
      {"FloatStorage", "float32"}, {"DoubleStorage", "float64"}, {"HalfStorage", "float16"},
      {"BFloat16Storage", "bfloat16"}, {"LongStorage", "int64"}, {"IntStorage", "int32"},
      {"ShortStorage", "int16"}, {"CharStorage", "int8"}, {"ByteStorage", "uint8"}, {"BoolStorage", "bool"}};
  for (auto& e : t)
    if (g.find(e.first) != std::string::npos) return e.second;
  return "unknown";
}
inline size_t dtype_size(const std::string& d) {
  if (d == "float32" || d == "int32") return 4;
  if (d == "float64" || d == "int64") return 8;
  if (d == "float16" || d == "bfloat16" || d == "int16") return 2;
  return 1;
}

inline VP reduce(const VP& fn, const VP& args) {
  if (fn->k == Val::GLOBAL) {
    const std::string& g = fn->s;
    if (g == "collections OrderedDict") return mk(Val::DICT);
    if (g == "torch._utils _rebuild_tensor_v2" && args->items.size() >= 4 && args->items[0]->k == Val::STORAGE &&
        args->items[1]->k == Val::INT) {
      VP t = mk(Val::TENSOR);
      t->s = args->items[0]->s;
      t->dtype = args->items[0]->dtype;
      t->offset = args->items[1]->i;
      for (auto& e : args->items[2]->items) { if (e->k != Val::INT) throw std::runtime_error("tensor shape entry is not an integer"); t->shape.push_back(e->i); }
      for (auto& e : args->items[3]->items) { if (e->k != Val::INT) throw std::runtime_error("tensor stride entry is not an integer"); t->stride.push_back(e->i); }
      return t;
    }
    if (g == "torch._utils _rebuild_parameter" && !args->items.empty()) return args->items[0];
  }
  VP o = mk(Val::OBJ);  // anything else (optimizer state objects, ...) is carried but never interpreted
  o->items = args->items;
  return o;
}

inline VP unpickle(const char* p, size_t n) {
  std::vector<VP> stack;
  std::vector<size_t> marks;
  std::map<uint32_t, VP> memo;
  size_t i = 0;
  auto need = [&](size_t k) { if (i + k > n) throw std::runtime_error("truncated pickle"); };
  auto pop = [&]() { if (stack.empty()) throw std::runtime_error("pickle stack underflow"); VP v = stack.back(); stack.pop_back(); return v; };
  auto top = [&]() -> VP& { if (stack.empty()) throw std::runtime_error("pickle stack underflow"); return stack.back(); };
  auto memo_at = [&](uint32_t k) -> VP { auto it = memo.find(k); if (it == memo.end()) throw std::runtime_error("pickle: unknown memo key"); return it->second; };
  auto pop_mark = [&]() {
    if (marks.empty()) throw std::runtime_error("pickle: no mark");
    const size_t m = marks.back();
    marks.pop_back();
    if (m > stack.size()) throw std::runtime_error("pickle: mark above the stack");
    std::vector<VP> v(stack.begin() + m, stack.end());
    stack.resize(m);
    return v;
  };
  auto line = [&]() { std::string s; while (true) { need(1); char c = p[i++]; if (c == '\n') break; s += c; } return s; };
  while (true) {
    need(1);
    const unsigned char op = (unsigned char)p[i++];
    switch (op) {
      case 0x80: need(1); ++i; break;                                   // PROTO
      case '}': stack.push_back(mk(Val::DICT)); break;
      case ']': stack.push_back(mk(Val::LIST)); break;
      case ')': stack.push_back(mk(Val::TUPLE)); break;
      case '(': marks.push_back(stack.size()); break;
      case 'N': stack.push_back(mk(Val::NONE)); break;
      case 0x88: case 0x89: { VP v = mk(Val::BOOL); v->i = op == 0x88; stack.push_back(v); break; }
      case 'K': { need(1); VP v = mk(Val::INT); v->i = (unsigned char)p[i]; i += 1; stack.push_back(v); break; }
      case 'M': { need(2); VP v = mk(Val::INT); v->i = rd16(p + i); i += 2; stack.push_back(v); break; }
      case 'J': { need(4); VP v = mk(Val::INT); v->i = (int32_t)rd32(p + i); i += 4; stack.push_back(v); break; }
      case 0x8a: { need(1); const int len = (unsigned char)p[i++]; need(len); int64_t x = 0;
                   for (int b = 0; b < len && b < 8; ++b) x |= (int64_t)(unsigned char)p[i + b] << (8 * b);
                   if (len > 0 && len < 8 && (p[i + len - 1] & 0x80)) x |= -((int64_t)1 << (8 * len));
                   i += len; VP v = mk(Val::INT); v->i = x; stack.push_back(v); break; }
      case 'G': { need(8); uint64_t b = 0; for (int k = 0; k < 8; ++k) b = (b << 8) | (unsigned char)p[i + k];
                  i += 8; VP v = mk(Val::FLOAT); memcpy(&v->d, &b, 8); stack.push_back(v); break; }
      case 'X': case 'T': { need(4); const uint32_t len = rd32(p + i); i += 4; need(len); VP v = mk(Val::STR);
                            v->s.assign(p + i, len); i += len; stack.push_back(v); break; }
      case 'U': case 0x8c: { need(1); const uint32_t len = (unsigned char)p[i++]; need(len); VP v = mk(Val::STR);
                             v->s.assign(p + i, len); i += len; stack.push_back(v); break; }
      case 'c': { VP v = mk(Val::GLOBAL); const std::string m = line(); v->s = m + " " + line(); stack.push_back(v); break; }
      case 'q': need(1); memo[(unsigned char)p[i]] = top(); i += 1; break;
      case 'r': need(4); memo[rd32(p + i)] = top(); i += 4; break;
      case 'h': need(1); stack.push_back(memo_at((unsigned char)p[i])); i += 1; break;
      case 'j': need(4); stack.push_back(memo_at(rd32(p + i))); i += 4; break;
      case 0x85: { VP t = mk(Val::TUPLE); t->items = {pop()}; stack.push_back(t); break; }
      case 0x86: { VP b = pop(), a = pop(); VP t = mk(Val::TUPLE); t->items = {a, b}; stack.push_back(t); break; }
      case 0x87: { VP c = pop(), b = pop(), a = pop(); VP t = mk(Val::TUPLE); t->items = {a, b, c}; stack.push_back(t); break; }
      case 't': { VP t = mk(Val::TUPLE); t->items = pop_mark(); stack.push_back(t); break; }
      case 'l': { VP t = mk(Val::LIST); t->items = pop_mark(); stack.push_back(t); break; }
      case 'a': { VP v = pop(); top()->items.push_back(v); break; }
      case 'e': { auto v = pop_mark(); for (auto& e : v) top()->items.push_back(e); break; }
      case 's': { VP v = pop(), k = pop(); top()->dict.push_back({k, v}); break; }
      case 'u': { auto v = pop_mark(); for (size_t k = 0; k + 1 < v.size(); k += 2) top()->dict.push_back({v[k], v[k + 1]}); break; }
      case 'Q': {  // BINPERSID: ('storage', StorageType, key, location, numel)
        VP pid = pop();
        VP s = mk(Val::STORAGE);
        if (pid->k == Val::TUPLE && pid->items.size() >= 3 && pid->items[0]->k == Val::STR && pid->items[0]->s == "storage") {
          s->dtype = storage_dtype(pid->items[1]->s);
          s->s = pid->items[2]->s;
        }
        stack.push_back(s);
        break;
      }
      case 'R': { VP args = pop(), fn = pop(); stack.push_back(reduce(fn, args)); break; }
      case 0x81: { VP args = pop(), cls = pop(); stack.push_back(reduce(cls, args)); break; }  // NEWOBJ
      case 'b': { VP state = pop(); (void)state; break; }  // BUILD: object state is never needed here
      case '.': return pop();
      default: {
        char msg[64];
        snprintf(msg, sizeof msg, "unsupported pickle opcode 0x%02x at %zu", op, i - 1);
        throw std::runtime_error(msg);
      }
    }
  }
}

}  // namespace detail

// Loads `path`; returns the tensors of ckpt['model_state_dict'] when that key exists
// (the reference trainer's layout), else of the top-level dict (flat layout).
inline Checkpoint load_checkpoint(const std::string& path) {
  using namespace detail;
  Checkpoint ck;
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("Failed to open file " + path);   // cpp/src/superpoint.cc:56-59
  ck.bytes.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
  auto dir = zip_directory(ck.bytes);
  std::string prefix;
  const ZipEntry* pkl = nullptr;
  for (auto& kv : dir) {
    const std::string& nm = kv.first;
    if (nm.size() >= 8 && nm.compare(nm.size() - 8, 8, "data.pkl") == 0) {
      pkl = &kv.second;
      prefix = nm.substr(0, nm.size() - 8);
    }
  }
  if (!pkl) throw std::runtime_error("no data.pkl in " + path + " (not a torch.save zip checkpoint)");
  VP root = unpickle(&ck.bytes[pkl->offset], pkl->size);
  if (root->k != Val::DICT) throw std::runtime_error("checkpoint root is not a dict");
  VP sd = root;
  for (auto& kv : root->dict)
    if (kv.first->k == Val::STR && kv.first->s == "model_state_dict" && kv.second->k == Val::DICT) sd = kv.second;
  for (auto& kv : sd->dict) {
    if (kv.first->k != Val::STR || kv.second->k != Val::TENSOR) continue;
    const Val& t = *kv.second;
    auto it = dir.find(prefix + "data/" + t.s);
    if (it == dir.end()) throw std::runtime_error("storage " + t.s + " of " + kv.first->s + " missing");
    Tensor out;
    out.dtype = t.dtype;
    out.shape = t.shape;
    // offset, shape and stride are pickled integers: refuse negative values and anything whose byte range, computed
    // without wrapping, leaves the storage entry
    if (t.stride.size() != t.shape.size()) throw std::runtime_error("tensor " + kv.first->s + ": stride rank differs from shape rank");
    if (t.offset < 0) throw std::runtime_error("tensor " + kv.first->s + ": negative storage offset");
    const size_t es = dtype_size(t.dtype);
    const uint64_t cap_elems = it->second.size / es;           // elements the storage entry holds
    uint64_t numel = 1;
    for (size_t d = t.shape.size(); d-- > 0;) {  // must be contiguous (state_dict tensors are)
      if (t.shape[d] < 0) throw std::runtime_error("tensor " + kv.first->s + ": negative dimension");
      if (t.shape[d] != 1 && t.stride[d] != (int64_t)numel) throw std::runtime_error("non-contiguous tensor " + kv.first->s);
      if (t.shape[d] != 0 && numel > cap_elems / (uint64_t)t.shape[d]) throw std::runtime_error("tensor out of storage: " + kv.first->s);
      numel *= (uint64_t)t.shape[d];
    }
    if ((uint64_t)t.offset > cap_elems || numel > cap_elems - (uint64_t)t.offset) throw std::runtime_error("tensor out of storage: " + kv.first->s);
    out.numel = (size_t)numel;
    out.data = &ck.bytes[it->second.offset + (size_t)t.offset * es];
    ck.tensors.push_back({kv.first->s, out});
  }
  return ck;
}

}  // namespace fpc_pt
