// torch_archive.cpp — see torch_archive.h.  Zip (stored entries, zip64 aware) + a closed pickle machine.
#include "torch_archive.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <map>
#include <memory>

#include "common.h"

namespace pcv {
namespace {

// ---------------------------------------------------------------- file mapping
struct Mapping {
    const uint8_t* p = nullptr;
    size_t n = 0;
    ~Mapping() {
        if (p) munmap(const_cast<uint8_t*>(p), n);
    }
};

void map_file(const std::string& path, Mapping& m) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) PCV_FAIL(PCV_ERR_IO, "cannot open %s", path.c_str());
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) {
        close(fd);
        PCV_FAIL(PCV_ERR_IO, "%s: empty or unreadable", path.c_str());
    }
    void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) PCV_FAIL(PCV_ERR_IO, "%s: mmap failed", path.c_str());
    m.p = (const uint8_t*)p;
    m.n = (size_t)st.st_size;
}

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

// ---------------------------------------------------------------- zip directory
struct Entry {
    const uint8_t* data;
    uint64_t size;
    bool stored;  // method 0; the code/ entries of a module archive are deflated and never read here
};

// name -> bytes of every entry; data.pkl and the storages must be method 0 (stored), which is what libtorch writes.
std::map<std::string, Entry> read_zip_directory(const Mapping& f, const std::string& path) {
    const uint8_t* p = f.p;
    const size_t n = f.n;
    if (n < 22) PCV_FAIL(PCV_ERR_IO, "%s: too short for a zip archive", path.c_str());
    // end-of-central-directory record: last 22 bytes + an up to 64 KiB comment
    size_t eocd = (size_t)-1;
    const size_t lowest = n > 22 + 65535 ? n - 22 - 65535 : 0;
    for (size_t i = n - 22 + 1; i-- > lowest;)
        if (rd32(p + i) == 0x06054b50u) {
            eocd = i;
            break;
        }
    if (eocd == (size_t)-1)
        PCV_FAIL(PCV_ERR_IO, "%s: not a zip archive (legacy non-zip torch files are not read; re-save the checkpoint)", path.c_str());
    uint64_t count = rd16(p + eocd + 10), cd_size = rd32(p + eocd + 12), cd_off = rd32(p + eocd + 16);
    if (count == 0xFFFFu || cd_size == 0xFFFFFFFFu || cd_off == 0xFFFFFFFFu) {  // zip64
        if (eocd < 20 || rd32(p + eocd - 20) != 0x07064b50u) PCV_FAIL(PCV_ERR_IO, "%s: zip64 locator missing", path.c_str());
        const uint64_t e64 = rd64(p + eocd - 20 + 8);
        if (e64 + 56 > n || rd32(p + e64) != 0x06064b50u) PCV_FAIL(PCV_ERR_IO, "%s: zip64 end record missing", path.c_str());
        count = rd64(p + e64 + 32);
        cd_size = rd64(p + e64 + 40);
        cd_off = rd64(p + e64 + 48);
    }
    if (cd_off > n || cd_size > n - cd_off) PCV_FAIL(PCV_ERR_IO, "%s: central directory out of range", path.c_str());
    std::map<std::string, Entry> dir;
    uint64_t q = cd_off;
    for (uint64_t e = 0; e < count; ++e) {
        if (q + 46 > cd_off + cd_size || rd32(p + q) != 0x02014b50u) PCV_FAIL(PCV_ERR_IO, "%s: damaged central directory", path.c_str());
        const uint16_t method = rd16(p + q + 10), nlen = rd16(p + q + 28), xlen = rd16(p + q + 30), clen = rd16(p + q + 32);
        uint64_t csize = rd32(p + q + 20), usize = rd32(p + q + 24), lho = rd32(p + q + 42);
        if (q + 46 + nlen + xlen + clen > cd_off + cd_size) PCV_FAIL(PCV_ERR_IO, "%s: damaged central directory", path.c_str());
        const std::string name((const char*)p + q + 46, nlen);
        // zip64 extended information: the 64-bit values of whichever fields were saturated, in this order
        const uint8_t* x = p + q + 46 + nlen;
        for (uint32_t o = 0; o + 4 <= xlen;) {
            const uint16_t id = rd16(x + o), sz = rd16(x + o + 2);
            if (o + 4 + sz > xlen) break;
            if (id == 0x0001) {
                uint32_t w = o + 4;
                const uint32_t wend = o + 4 + sz;
                if (usize == 0xFFFFFFFFu && w + 8 <= wend) usize = rd64(x + w), w += 8;
                if (csize == 0xFFFFFFFFu && w + 8 <= wend) csize = rd64(x + w), w += 8;
                if (lho == 0xFFFFFFFFu && w + 8 <= wend) lho = rd64(x + w), w += 8;
            }
            o += 4u + sz;
        }
        q += 46u + nlen + xlen + clen;
        if (!name.empty() && name.back() == '/') continue;  // directory entry
        if (lho + 30 > n || rd32(p + lho) != 0x04034b50u) PCV_FAIL(PCV_ERR_IO, "%s: entry %s has no local header", path.c_str(), name.c_str());
        const uint64_t start = lho + 30u + rd16(p + lho + 26) + rd16(p + lho + 28);
        if (start > n || csize > n - start) PCV_FAIL(PCV_ERR_IO, "%s: entry %s runs past the end of the file", path.c_str(), name.c_str());
        dir[name] = Entry{p + start, csize, method == 0 && csize == usize};
    }
    return dir;
}

// ---------------------------------------------------------------- pickle values
struct Val;
using Ref = std::shared_ptr<Val>;

struct Val {
    enum Kind { None, Bool, Int, Float, Str, Tuple, List, Dict, Global, Storage, Tensor, Object, Mark } kind = None;
    int64_t i = 0;
    double f = 0;
    std::string s;           // Str; Global: "module name"; Storage: entry key
    std::string s2;          // Storage: storage class
    std::vector<Ref> items;  // Tuple / List; Dict and Object state: key, value, key, value ...
    // Tensor
    Ref storage;
    int64_t offset = 0;
    std::vector<int64_t> shape, stride;
};

Ref mk(Val::Kind k) {
    auto v = std::make_shared<Val>();
    v->kind = k;
    return v;
}

struct Machine {
    const std::string& path;
    const uint8_t* p;
    size_t n, pc = 0;
    std::vector<Ref> stack;
    std::map<uint32_t, Ref> memo;

    [[noreturn]] void bad(const char* what) { PCV_FAIL(PCV_ERR_IO, "%s: data.pkl: %s at byte %zu", path.c_str(), what, pc); }
    void need(size_t k) {
        if (pc + k > n) bad("truncated pickle");
    }
    Ref pop() {
        if (stack.empty()) bad("stack underflow");
        Ref v = stack.back();
        stack.pop_back();
        return v;
    }
    std::vector<Ref> pop_to_mark() {
        size_t m = stack.size();
        while (m > 0 && stack[m - 1]->kind != Val::Mark) --m;
        if (m == 0) bad("no MARK on the stack");
        std::vector<Ref> out(stack.begin() + (long)m, stack.end());
        stack.resize(m - 1);
        return out;
    }
    std::string line() {  // newline-terminated text argument
        size_t e = pc;
        while (e < n && p[e] != '\n') ++e;
        if (e >= n) bad("unterminated line");
        std::string s((const char*)p + pc, e - pc);
        pc = e + 1;
        return s;
    }
    void push_str(size_t len) {
        need(len);
        Ref v = mk(Val::Str);
        v->s.assign((const char*)p + pc, len);
        pc += len;
        stack.push_back(v);
    }
    static std::vector<int64_t> int_list(const Ref& v, Machine& mch) {
        if (v->kind != Val::Tuple && v->kind != Val::List) mch.bad("expected a tuple of integers");
        std::vector<int64_t> out;
        for (const Ref& e : v->items) {
            if (e->kind != Val::Int) mch.bad("expected a tuple of integers");
            out.push_back(e->i);
        }
        return out;
    }

    // the constructors the two writers emit; everything else is refused
    Ref reduce(const Ref& fn, const Ref& args) {
        if (fn->kind != Val::Global || args->kind != Val::Tuple) bad("REDUCE of something that is not a known constructor");
        const std::string& g = fn->s;
        const auto& a = args->items;
        if (g == "collections OrderedDict") return mk(Val::Dict);
        if (g == "torch._utils _rebuild_tensor_v2" || g == "torch._utils _rebuild_tensor") {
            if (a.size() < 4 || a[0]->kind != Val::Storage || a[1]->kind != Val::Int) bad("malformed tensor record");
            Ref t = mk(Val::Tensor);
            t->storage = a[0];
            t->offset = a[1]->i;
            t->shape = int_list(a[2], *this);
            t->stride = int_list(a[3], *this);
            if (t->shape.size() != t->stride.size()) bad("tensor sizes and strides differ in rank");
            return t;
        }
        if (g == "torch._utils _rebuild_parameter" || g == "torch._utils _rebuild_parameter_with_state") {
            if (a.empty() || a[0]->kind != Val::Tensor) bad("malformed parameter record");
            return a[0];
        }
        PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: data.pkl calls %s, which this reader does not know (nothing from the file is executed)",
                 path.c_str(), g.c_str());
    }

    Ref run() {
        for (;;) {
            need(1);
            const uint8_t op = p[pc++];
            switch (op) {
                case 0x80: need(1); ++pc; break;                      // PROTO
                case 0x95: need(8); pc += 8; break;                   // FRAME
                case '.': return pop();                               // STOP
                case '(': stack.push_back(mk(Val::Mark)); break;
                case 'N': stack.push_back(mk(Val::None)); break;
                case 0x88: case 0x89: { Ref v = mk(Val::Bool); v->i = op == 0x88; stack.push_back(v); break; }
                case 'K': { need(1); Ref v = mk(Val::Int); v->i = p[pc]; pc += 1; stack.push_back(v); break; }
                case 'M': { need(2); Ref v = mk(Val::Int); v->i = rd16(p + pc); pc += 2; stack.push_back(v); break; }
                case 'J': { need(4); Ref v = mk(Val::Int); v->i = (int32_t)rd32(p + pc); pc += 4; stack.push_back(v); break; }
                case 0x8a: case 0x8b: {                                // LONG1 / LONG4
                    size_t len;
                    if (op == 0x8a) { need(1); len = p[pc++]; } else { need(4); len = rd32(p + pc); pc += 4; }
                    need(len);
                    if (len > 8) bad("integer wider than 64 bits");
                    uint64_t u = 0;
                    for (size_t k = 0; k < len; ++k) u |= (uint64_t)p[pc + k] << (8 * k);
                    if (len > 0 && len < 8 && (p[pc + len - 1] & 0x80)) u |= ~0ull << (8 * len);  // sign extension
                    pc += len;
                    Ref v = mk(Val::Int); v->i = (int64_t)u; stack.push_back(v);
                    break;
                }
                case 'G': {                                            // BINFLOAT, big-endian
                    need(8);
                    uint64_t u = 0;
                    for (int k = 0; k < 8; ++k) u = (u << 8) | p[pc + k];
                    pc += 8;
                    Ref v = mk(Val::Float); std::memcpy(&v->f, &u, 8); stack.push_back(v);
                    break;
                }
                case 'X': case 'T': case 'B': { need(4); const size_t len = rd32(p + pc); pc += 4; push_str(len); break; }  // BINUNICODE / BINSTRING / BINBYTES
                case 0x8c: case 'U': case 'C': { need(1); const size_t len = p[pc++]; push_str(len); break; }               // SHORT_*
                case 0x8d: case 0x8e: { need(8); const uint64_t len = rd64(p + pc); pc += 8; if (len > n) bad("string too long"); push_str((size_t)len); break; }
                case 'c': {                                            // GLOBAL "module\nname\n"
                    Ref v = mk(Val::Global);
                    const std::string mod = line();
                    v->s = mod + " " + line();
                    stack.push_back(v);
                    break;
                }
                case 0x93: {                                           // STACK_GLOBAL
                    Ref name = pop(), mod = pop();
                    if (name->kind != Val::Str || mod->kind != Val::Str) bad("STACK_GLOBAL without strings");
                    Ref v = mk(Val::Global); v->s = mod->s + " " + name->s; stack.push_back(v);
                    break;
                }
                case ')': stack.push_back(mk(Val::Tuple)); break;
                case '}': stack.push_back(mk(Val::Dict)); break;
                case ']': stack.push_back(mk(Val::List)); break;
                case 't': { Ref v = mk(Val::Tuple); v->items = pop_to_mark(); stack.push_back(v); break; }
                case 0x85: case 0x86: case 0x87: {
                    const size_t k = (size_t)(op - 0x84);
                    if (stack.size() < k) bad("stack underflow");
                    Ref v = mk(Val::Tuple);
                    v->items.assign(stack.end() - (long)k, stack.end());
                    stack.resize(stack.size() - k);
                    stack.push_back(v);
                    break;
                }
                case 'q': { need(1); if (stack.empty()) bad("BINPUT on an empty stack"); memo[p[pc]] = stack.back(); pc += 1; break; }
                case 'r': { need(4); if (stack.empty()) bad("LONG_BINPUT on an empty stack"); memo[rd32(p + pc)] = stack.back(); pc += 4; break; }
                case 0x94: { if (stack.empty()) bad("MEMOIZE on an empty stack"); const uint32_t k = (uint32_t)memo.size(); memo[k] = stack.back(); break; }
                case 'h': case 'j': {
                    uint32_t k;
                    if (op == 'h') { need(1); k = p[pc]; pc += 1; } else { need(4); k = rd32(p + pc); pc += 4; }
                    auto it = memo.find(k);
                    if (it == memo.end()) bad("memo key never stored");
                    stack.push_back(it->second);
                    break;
                }
                case 'Q': {                                            // BINPERSID: ('storage', <class>, key, location, numel)
                    Ref id = pop();
                    if (id->kind != Val::Tuple || id->items.size() < 5 || id->items[0]->kind != Val::Str || id->items[0]->s != "storage" ||
                        id->items[1]->kind != Val::Global || id->items[2]->kind != Val::Str || id->items[4]->kind != Val::Int)
                        bad("persistent id is not a storage record");
                    Ref v = mk(Val::Storage);
                    v->s = id->items[2]->s;
                    const std::string& cls = id->items[1]->s;
                    v->s2 = cls.substr(cls.find(' ') == std::string::npos ? 0 : cls.find(' ') + 1);
                    v->i = id->items[4]->i;
                    stack.push_back(v);
                    break;
                }
                case 'R': { Ref args = pop(), fn = pop(); stack.push_back(reduce(fn, args)); break; }
                case 0x81: {                                           // NEWOBJ: the module object of an OutputArchive
                    Ref args = pop(), cls = pop();
                    if (cls->kind != Val::Global) bad("NEWOBJ of something that is not a class");
                    if (cls->s == "collections OrderedDict") { stack.push_back(mk(Val::Dict)); break; }
                    if (cls->s.compare(0, 9, "__torch__") != 0)
                        PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: data.pkl instantiates %s, which this reader does not know", path.c_str(), cls->s.c_str());
                    Ref v = mk(Val::Object); v->s = cls->s; stack.push_back(v);
                    break;
                }
                case 'b': {                                            // BUILD
                    Ref state = pop();
                    if (stack.empty()) bad("BUILD on an empty stack");
                    Ref& obj = stack.back();
                    if (obj->kind == Val::Object) {
                        if (state->kind != Val::Dict) bad("module state is not a dict");
                        obj->items = state->items;
                    }  // an OrderedDict's _metadata: nothing to keep
                    break;
                }
                case 's': {
                    Ref v = pop(), k = pop();
                    if (stack.empty() || stack.back()->kind != Val::Dict) bad("SETITEM without a dict");
                    stack.back()->items.push_back(k);
                    stack.back()->items.push_back(v);
                    break;
                }
                case 'u': {
                    std::vector<Ref> kv = pop_to_mark();
                    if (stack.empty() || stack.back()->kind != Val::Dict || (kv.size() & 1)) bad("SETITEMS without a dict");
                    for (Ref& e : kv) stack.back()->items.push_back(e);
                    break;
                }
                case 'a': {
                    Ref v = pop();
                    if (stack.empty() || stack.back()->kind != Val::List) bad("APPEND without a list");
                    stack.back()->items.push_back(v);
                    break;
                }
                case 'e': {
                    std::vector<Ref> vs = pop_to_mark();
                    if (stack.empty() || stack.back()->kind != Val::List) bad("APPENDS without a list");
                    for (Ref& e : vs) stack.back()->items.push_back(e);
                    break;
                }
                default: {
                    char msg[64];
                    std::snprintf(msg, sizeof msg, "opcode 0x%02x is not supported", op);
                    --pc;
                    bad(msg);
                }
            }
        }
    }
};

float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else {  // subnormal
            int e = -1;
            uint32_t mm = man;
            do {
                ++e;
                mm <<= 1;
            } while (!(mm & 0x400u));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mm & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7f800000u | (man << 13);
    } else {
        u = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

struct StorageKind {
    int dtype;
    int esize;
};

StorageKind storage_kind(const std::string& cls) {
    if (cls == "FloatStorage") return {AR_F32, 4};
    if (cls == "HalfStorage") return {AR_F16, 2};
    if (cls == "BFloat16Storage") return {AR_BF16, 2};
    if (cls == "DoubleStorage") return {AR_F64, 8};
    return {AR_OTHER, 0};
}

inline float element(const uint8_t* base, int dtype, int64_t idx) {
    switch (dtype) {
        case AR_F32: {
            float f;
            std::memcpy(&f, base + 4 * idx, 4);
            return f;
        }
        case AR_F16: return half_bits_to_float(rd16(base + 2 * idx));
        case AR_BF16: {
            const uint32_t u = (uint32_t)rd16(base + 2 * idx) << 16;
            float f;
            std::memcpy(&f, &u, 4);
            return f;
        }
        default: {
            double d;
            std::memcpy(&d, base + 8 * idx, 8);
            return (float)d;
        }
    }
}

}  // namespace

void read_torch_archive(const std::string& path, const std::function<bool(const std::string&)>& want,
                        const std::function<void(const ArchiveTensor&)>& fn) {
    Mapping file;
    map_file(path, file);
    const std::map<std::string, Entry> dir = read_zip_directory(file, path);
    // <root>/data.pkl; the root is the archive's stem at save time ("rust_model", "archive", ...)
    std::string root;
    const Entry* pkl = nullptr;
    for (const auto& kv : dir) {
        const std::string& nm = kv.first;
        if (nm == "data.pkl" || (nm.size() > 9 && nm.compare(nm.size() - 9, 9, "/data.pkl") == 0)) {
            if (pkl && nm.size() >= root.size() + 8) continue;  // keep the shallowest
            pkl = &kv.second;
            root = nm.substr(0, nm.size() - 8);  // "" or "<root>/"
        }
    }
    if (!pkl) PCV_FAIL(PCV_ERR_IO, "%s: no data.pkl in the archive (not a torch checkpoint)", path.c_str());
    if (!pkl->stored) PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: data.pkl is compressed; libtorch stores it uncompressed", path.c_str());
    auto bo = dir.find(root + "byteorder");
    if (bo != dir.end() && !(bo->second.size >= 6 && std::memcmp(bo->second.data, "little", 6) == 0))
        PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: big-endian archive", path.c_str());

    Machine mch{path, pkl->data, (size_t)pkl->size};
    Ref top = mch.run();
    // module object (OutputArchive) or state dict (torch.save), possibly wrapped as {"state_dict": {...}}
    if (top->kind == Val::Dict)
        for (size_t i = 0; i + 1 < top->items.size(); i += 2)
            if (top->items[i]->kind == Val::Str && top->items[i]->s == "state_dict" && top->items[i + 1]->kind == Val::Dict) {
                top = top->items[i + 1];
                break;
            }
    if (top->kind != Val::Object && top->kind != Val::Dict) PCV_FAIL(PCV_ERR_IO, "%s: data.pkl holds neither a module nor a state dict", path.c_str());

    ArchiveTensor out;
    for (size_t i = 0; i + 1 < top->items.size(); i += 2) {
        const Ref& key = top->items[i];
        const Ref& val = top->items[i + 1];
        if (key->kind != Val::Str || val->kind != Val::Tensor) continue;  // "training" flags, nested modules: not weights
        if (!want(key->s)) continue;
        const Val& st = *val->storage;
        const StorageKind sk = storage_kind(st.s2);
        out.name = key->s;
        out.dtype = sk.dtype;
        out.dtype_name = st.s2;
        out.shape = val->shape;
        out.numel = 1;
        for (int64_t d : val->shape) {  // (values straight out of the pickle: every product is checked)
            if (d < 0 || d > (int64_t)1 << 40 || __builtin_mul_overflow(out.numel, d, &out.numel) || out.numel > (int64_t)1 << 40)
                PCV_FAIL(PCV_ERR_IO, "%s: tensor %s has an implausible shape", path.c_str(), key->s.c_str());
        }
        if (val->stride.size() != val->shape.size()) PCV_FAIL(PCV_ERR_IO, "%s: tensor %s has %zu strides for %zu dimensions", path.c_str(), key->s.c_str(), val->stride.size(), val->shape.size());
        out.values.clear();
        if (sk.dtype != AR_OTHER && out.numel > 0) {
            auto e = dir.find(root + "data/" + st.s);
            if (e == dir.end()) PCV_FAIL(PCV_ERR_IO, "%s: storage %s of tensor %s is missing", path.c_str(), st.s.c_str(), key->s.c_str());
            if (!e->second.stored) PCV_FAIL(PCV_ERR_UNSUPPORTED, "%s: storage %s is compressed; libtorch stores tensors uncompressed", path.c_str(), st.s.c_str());
            const int64_t avail = (int64_t)(e->second.size / (uint64_t)sk.esize);
            // A weight is a view INTO its storage: no more elements than the storage holds (a broadcast view — stride 0
            // over a long dimension — would have this reader allocate what the file never contained), offset and every
            // stride inside it, and the furthest element the view touches — summed with overflow checks — as well.
            if (out.numel > avail) PCV_FAIL(PCV_ERR_IO, "%s: tensor %s has more elements than its storage", path.c_str(), key->s.c_str());
            if (val->offset < 0 || val->offset >= avail) PCV_FAIL(PCV_ERR_IO, "%s: tensor %s starts outside its storage", path.c_str(), key->s.c_str());
            int64_t last = val->offset;
            bool contiguous = true;
            int64_t expect = 1;
            for (size_t d = val->shape.size(); d-- > 0;) {
                const int64_t st_d = val->stride[d], n_d = val->shape[d];  // n_d >= 1: numel > 0
                if (st_d < 0 || st_d > avail) PCV_FAIL(PCV_ERR_IO, "%s: tensor %s has a stride outside its storage", path.c_str(), key->s.c_str());
                int64_t reach = 0;
                if (__builtin_mul_overflow(n_d - 1, st_d, &reach) || __builtin_add_overflow(last, reach, &last) || last >= avail)
                    PCV_FAIL(PCV_ERR_IO, "%s: tensor %s reaches past its storage", path.c_str(), key->s.c_str());
                if (n_d != 1 && st_d != expect) contiguous = false;
                expect *= n_d;  // <= numel
            }
            out.values.resize((size_t)out.numel);
            const uint8_t* base = e->second.data;
            if (contiguous) {
                if (sk.dtype == AR_F32) {
                    std::memcpy(out.values.data(), base + 4 * val->offset, (size_t)out.numel * 4);  // little-endian host
                } else {
                    for (int64_t k = 0; k < out.numel; ++k) out.values[(size_t)k] = element(base, sk.dtype, val->offset + k);
                }
            } else {  // a view (transposed weight ...): walk the index space
                const size_t R = val->shape.size();
                std::vector<int64_t> idx(R, 0);
                for (int64_t k = 0; k < out.numel; ++k) {
                    int64_t src = val->offset;
                    for (size_t d = 0; d < R; ++d) src += idx[d] * val->stride[d];
                    out.values[(size_t)k] = element(base, sk.dtype, src);
                    for (size_t d = R; d-- > 0;) {
                        if (++idx[d] < val->shape[d]) break;
                        idx[d] = 0;
                    }
                }
            }
        }
        fn(out);
    }
}

}  // namespace pcv
