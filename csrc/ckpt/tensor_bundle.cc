#include "tensor_bundle.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace dtf {
namespace ckpt {

// ------------------------------------------------------------------------------------------
// CRC32C (Castagnoli), table driven, plus the LevelDB/TensorFlow mask.
namespace {
struct CrcTable {
  uint32_t t[8][256];
  CrcTable() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
  }
};
const CrcTable& crc_table() {
  static CrcTable tab;
  return tab;
}
}  // namespace

uint32_t crc32c(const void* data, size_t n, uint32_t init) {
  const CrcTable& T = crc_table();
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = init ^ 0xffffffffu;
  while (n >= 8) {
    uint32_t lo, hi;
    std::memcpy(&lo, p, 4);
    std::memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^
        T.t[3][hi & 0xff] ^ T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

uint32_t crc32c_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }
uint32_t crc32c_unmask(uint32_t masked) {
  uint32_t rot = masked - 0xa282ead8u;
  return (rot >> 17) | (rot << 15);
}

// ------------------------------------------------------------------------------------------
// varint / fixed / protobuf helpers
namespace {

void put_varint64(std::string* dst, uint64_t v) {
  while (v >= 0x80) {
    dst->push_back(static_cast<char>(v | 0x80));
    v >>= 7;
  }
  dst->push_back(static_cast<char>(v));
}
void put_fixed32(std::string* dst, uint32_t v) {
  char b[4] = {char(v & 0xff), char((v >> 8) & 0xff), char((v >> 16) & 0xff), char((v >> 24) & 0xff)};
  dst->append(b, 4);
}
bool get_varint64(const std::string& s, size_t* pos, uint64_t* out) {
  uint64_t r = 0;
  for (int shift = 0; shift <= 63 && *pos < s.size(); shift += 7) {
    uint8_t b = static_cast<uint8_t>(s[(*pos)++]);
    r |= uint64_t(b & 0x7f) << shift;
    if (!(b & 0x80)) {
      *out = r;
      return true;
    }
  }
  return false;
}
uint32_t get_fixed32(const std::string& s, size_t pos) {
  if (pos + 4 > s.size()) throw std::runtime_error("tensor bundle: truncated fixed32");
  return uint32_t(uint8_t(s[pos])) | (uint32_t(uint8_t(s[pos + 1])) << 8) | (uint32_t(uint8_t(s[pos + 2])) << 16) |
         (uint32_t(uint8_t(s[pos + 3])) << 24);
}

void pb_varint_field(std::string* dst, int field, uint64_t v) {
  put_varint64(dst, (uint64_t(field) << 3) | 0);
  put_varint64(dst, v);
}
void pb_bytes_field(std::string* dst, int field, const std::string& v) {
  put_varint64(dst, (uint64_t(field) << 3) | 2);
  put_varint64(dst, v.size());
  dst->append(v);
}
void pb_fixed32_field(std::string* dst, int field, uint32_t v) {
  put_varint64(dst, (uint64_t(field) << 3) | 5);
  put_fixed32(dst, v);
}

std::string encode_header(int num_shards) {
  std::string s;
  pb_varint_field(&s, 1, num_shards);  // num_shards
  // endianness LITTLE = 0 is the proto3 default and is not emitted
  std::string version;
  pb_varint_field(&version, 1, 1);     // VersionDef.producer = 1
  pb_bytes_field(&s, 3, version);
  return s;
}

std::string encode_entry(const TensorEntry& e) {
  std::string s;
  pb_varint_field(&s, 1, e.dtype);
  std::string shape;
  for (int64_t d : e.shape) {
    std::string dim;
    if (d != 0) pb_varint_field(&dim, 1, static_cast<uint64_t>(d));
    pb_bytes_field(&shape, 2, dim);
  }
  pb_bytes_field(&s, 2, shape);
  // shard_id 0 omitted
  if (e.offset != 0) pb_varint_field(&s, 4, static_cast<uint64_t>(e.offset));
  if (e.size != 0) pb_varint_field(&s, 5, static_cast<uint64_t>(e.size));
  if (e.crc != 0) pb_fixed32_field(&s, 6, e.crc);
  return s;
}

// Minimal protobuf walker: calls fn(field, wiretype, varint value, bytes value).
template <typename Fn>
void pb_walk(const std::string& s, Fn fn) {
  size_t pos = 0;
  while (pos < s.size()) {
    uint64_t tag;
    if (!get_varint64(s, &pos, &tag)) throw std::runtime_error("tensor bundle: bad proto tag");
    const int field = int(tag >> 3), wt = int(tag & 7);
    uint64_t v = 0;
    std::string bytes;
    if (wt == 0) {
      if (!get_varint64(s, &pos, &v)) throw std::runtime_error("tensor bundle: bad varint");
    } else if (wt == 2) {
      uint64_t n;
      if (!get_varint64(s, &pos, &n) || pos + n > s.size()) throw std::runtime_error("tensor bundle: bad length");
      bytes = s.substr(pos, n);
      pos += n;
    } else if (wt == 5) {
      v = get_fixed32(s, pos);
      pos += 4;
    } else if (wt == 1) {
      if (pos + 8 > s.size()) throw std::runtime_error("tensor bundle: truncated fixed64");
      pos += 8;
    } else {
      throw std::runtime_error("tensor bundle: unsupported wire type");
    }
    fn(field, wt, v, bytes);
  }
}

TensorEntry decode_entry(const std::string& s) {
  TensorEntry e;
  e.dtype = 0;
  pb_walk(s, [&](int field, int, uint64_t v, const std::string& bytes) {
    switch (field) {
      case 1: e.dtype = int(v); break;
      case 2:
        pb_walk(bytes, [&](int f2, int, uint64_t, const std::string& dimbytes) {
          if (f2 != 2) return;
          int64_t size = 0;
          pb_walk(dimbytes, [&](int f3, int, uint64_t v3, const std::string&) {
            if (f3 == 1) size = int64_t(v3);
          });
          e.shape.push_back(size);
        });
        break;
      case 4: e.offset = int64_t(v); break;
      case 5: e.size = int64_t(v); break;
      case 6: e.crc = uint32_t(v); break;
      default: break;
    }
  });
  return e;
}

// ------------------------------------------------------------------------------------------
// LevelDB table blocks
class BlockBuilder {
 public:
  explicit BlockBuilder(int restart_interval) : interval_(restart_interval) { restarts_.push_back(0); }
  void add(const std::string& key, const std::string& value) {
    size_t shared = 0;
    if (counter_ < interval_) {
      const size_t m = std::min(last_key_.size(), key.size());
      while (shared < m && last_key_[shared] == key[shared]) ++shared;
    } else {
      restarts_.push_back(uint32_t(buf_.size()));
      counter_ = 0;
    }
    put_varint64(&buf_, shared);
    put_varint64(&buf_, key.size() - shared);
    put_varint64(&buf_, value.size());
    buf_.append(key, shared, std::string::npos);
    buf_.append(value);
    last_key_ = key;
    ++counter_;
  }
  std::string finish() {
    std::string out = buf_;
    for (uint32_t r : restarts_) put_fixed32(&out, r);
    put_fixed32(&out, uint32_t(restarts_.size()));
    return out;
  }
  size_t estimate() const { return buf_.size() + restarts_.size() * 4 + 4; }
  bool empty() const { return buf_.empty(); }
  void reset() {
    buf_.clear();
    restarts_.assign(1, 0);
    counter_ = 0;
    last_key_.clear();
  }

 private:
  int interval_;
  int counter_ = 0;
  std::string buf_, last_key_;
  std::vector<uint32_t> restarts_;
};

void shortest_separator(std::string* start, const std::string& limit) {
  const size_t m = std::min(start->size(), limit.size());
  size_t d = 0;
  while (d < m && (*start)[d] == limit[d]) ++d;
  if (d >= m) return;  // one is a prefix of the other
  const uint8_t b = uint8_t((*start)[d]);
  if (b < 0xff && b + 1 < uint8_t(limit[d])) {
    (*start)[d] = char(b + 1);
    start->resize(d + 1);
  }
}
void short_successor(std::string* key) {
  for (size_t i = 0; i < key->size(); ++i) {
    const uint8_t b = uint8_t((*key)[i]);
    if (b != 0xff) {
      (*key)[i] = char(b + 1);
      key->resize(i + 1);
      return;
    }
  }
}

struct Handle {
  uint64_t offset = 0, size = 0;
};
void encode_handle(std::string* dst, const Handle& h) {
  put_varint64(dst, h.offset);
  put_varint64(dst, h.size);
}

Handle write_block(std::string* file, const std::string& contents) {
  Handle h;
  h.offset = file->size();
  h.size = contents.size();
  file->append(contents);
  const char type = 0;  // kNoCompression
  uint32_t crc = crc32c(contents.data(), contents.size());
  crc = crc32c(&type, 1, crc);
  file->push_back(type);
  put_fixed32(file, crc32c_mask(crc));
  return h;
}

const uint64_t kTableMagic = 0xdb4775248b80fb57ull;

std::vector<std::pair<std::string, std::string>> parse_block(const std::string& file, const Handle& h, bool verify) {
  if (h.offset + h.size + 5 > file.size()) throw std::runtime_error("tensor bundle: block out of range");
  const std::string contents = file.substr(h.offset, h.size);
  const char type = file[h.offset + h.size];
  if (type != 0) throw std::runtime_error("tensor bundle: compressed index blocks are not supported");
  if (verify) {
    uint32_t crc = crc32c(contents.data(), contents.size());
    crc = crc32c(&type, 1, crc);
    if (crc32c_mask(crc) != get_fixed32(file, h.offset + h.size + 1))
      throw std::runtime_error("tensor bundle: block checksum mismatch");
  }
  if (contents.size() < 4) throw std::runtime_error("tensor bundle: short block");
  const uint32_t n_restarts = get_fixed32(contents, contents.size() - 4);
  const size_t data_end = contents.size() - 4 - size_t(n_restarts) * 4;
  std::vector<std::pair<std::string, std::string>> out;
  size_t pos = 0;
  std::string key;
  while (pos < data_end) {
    uint64_t shared, non_shared, vlen;
    if (!get_varint64(contents, &pos, &shared) || !get_varint64(contents, &pos, &non_shared) ||
        !get_varint64(contents, &pos, &vlen) || shared > key.size() || pos + non_shared + vlen > data_end)
      throw std::runtime_error("tensor bundle: corrupt block entry");
    key.resize(shared);
    key.append(contents, pos, non_shared);
    pos += non_shared;
    out.emplace_back(key, contents.substr(pos, vlen));
    pos += vlen;
  }
  return out;
}

std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
void write_file_atomic(const std::string& path, const std::string& bytes) {
  const std::string tmp = path + ".tempstate";
  {
    std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("cannot write " + tmp);
    f.write(bytes.data(), std::streamsize(bytes.size()));
    if (!f) throw std::runtime_error("short write to " + tmp);
  }
  if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("rename failed for " + path);
}

}  // namespace

std::string build_table(const std::vector<std::pair<std::string, std::string>>& kv, size_t block_size,
                        int restart_interval) {
  std::string file;
  BlockBuilder data(restart_interval), index(1);
  std::string last_key;
  bool pending = false, have_last = false;
  Handle pending_handle;
  std::string prev;
  for (const auto& e : kv) {
    if (have_last && !(prev < e.first)) throw std::runtime_error("build_table: keys must be strictly increasing");
    prev = e.first;
    have_last = true;
    if (pending) {
      shortest_separator(&last_key, e.first);
      std::string h;
      encode_handle(&h, pending_handle);
      index.add(last_key, h);
      pending = false;
    }
    data.add(e.first, e.second);
    last_key = e.first;
    if (data.estimate() >= block_size) {
      pending_handle = write_block(&file, data.finish());
      data.reset();
      pending = true;
    }
  }
  if (!data.empty()) {
    pending_handle = write_block(&file, data.finish());
    data.reset();
    pending = true;
  }
  // metaindex block (no filter policy: empty)
  BlockBuilder meta(restart_interval);
  const Handle meta_handle = write_block(&file, meta.finish());
  if (pending) {
    short_successor(&last_key);
    std::string h;
    encode_handle(&h, pending_handle);
    index.add(last_key, h);
  }
  const Handle index_handle = write_block(&file, index.finish());
  std::string footer;
  encode_handle(&footer, meta_handle);
  encode_handle(&footer, index_handle);
  footer.resize(40, '\0');
  put_fixed32(&footer, uint32_t(kTableMagic & 0xffffffffu));
  put_fixed32(&footer, uint32_t(kTableMagic >> 32));
  file.append(footer);
  return file;
}

std::vector<std::pair<std::string, std::string>> parse_table(const std::string& file) {
  if (file.size() < 48) throw std::runtime_error("tensor bundle: index file too short");
  const size_t foot = file.size() - 48;
  const uint64_t magic = uint64_t(get_fixed32(file, foot + 40)) | (uint64_t(get_fixed32(file, foot + 44)) << 32);
  if (magic != kTableMagic) throw std::runtime_error("tensor bundle: bad table magic");
  size_t pos = foot;
  Handle meta, index;
  if (!get_varint64(file, &pos, &meta.offset) || !get_varint64(file, &pos, &meta.size) ||
      !get_varint64(file, &pos, &index.offset) || !get_varint64(file, &pos, &index.size))
    throw std::runtime_error("tensor bundle: bad footer");
  std::vector<std::pair<std::string, std::string>> out;
  for (const auto& ie : parse_block(file, index, true)) {
    Handle h;
    size_t p = 0;
    if (!get_varint64(ie.second, &p, &h.offset) || !get_varint64(ie.second, &p, &h.size))
      throw std::runtime_error("tensor bundle: bad index entry");
    for (auto& kv : parse_block(file, h, true)) out.push_back(std::move(kv));
  }
  return out;
}

// ------------------------------------------------------------------------------------------
BundleWriter::BundleWriter(const std::string& prefix) : prefix_(prefix) {}

void BundleWriter::add(const std::string& name, int dtype, const std::vector<int64_t>& shape, const void* data,
                       size_t bytes) {
  if (finished_) throw std::runtime_error("BundleWriter: already finished");
  if (name.empty()) throw std::runtime_error("BundleWriter: empty tensor name");
  if (entries_.count(name)) throw std::runtime_error("BundleWriter: duplicate tensor " + name);
  TensorEntry e;
  e.dtype = dtype;
  e.shape = shape;
  e.offset = int64_t(data_.size());
  e.size = int64_t(bytes);
  e.crc = crc32c_mask(crc32c(data, bytes));
  data_.append(static_cast<const char*>(data), bytes);
  entries_[name] = e;
}

void BundleWriter::finish() {
  if (finished_) return;
  std::vector<std::pair<std::string, std::string>> kv;
  kv.emplace_back("", encode_header(1));
  for (const auto& e : entries_) kv.emplace_back(e.first, encode_entry(e.second));
  write_file_atomic(prefix_ + ".data-00000-of-00001", data_);
  write_file_atomic(prefix_ + ".index", build_table(kv));
  finished_ = true;
}

BundleReader::BundleReader(const std::string& prefix) : prefix_(prefix) {
  const auto kv = parse_table(read_file(prefix + ".index"));
  bool have_header = false;
  for (const auto& e : kv) {
    if (e.first.empty()) {
      pb_walk(e.second, [&](int field, int, uint64_t v, const std::string&) {
        if (field == 1) num_shards_ = int(v);
        if (field == 2 && v != 0) throw std::runtime_error("tensor bundle: big-endian bundles are not supported");
      });
      have_header = true;
    } else {
      entries_[e.first] = decode_entry(e.second);
    }
  }
  if (!have_header) throw std::runtime_error("tensor bundle: missing header entry");
  if (num_shards_ != 1) throw std::runtime_error("tensor bundle: only single-shard bundles are supported");
}

std::string BundleReader::read(const std::string& name) const {
  auto it = entries_.find(name);
  if (it == entries_.end()) throw std::runtime_error("tensor bundle: no tensor named " + name);
  const TensorEntry& e = it->second;
  std::ifstream f(prefix_ + ".data-00000-of-00001", std::ios::binary);
  if (!f) throw std::runtime_error("cannot open data shard of " + prefix_);
  f.seekg(e.offset);
  std::string buf(size_t(e.size), '\0');
  f.read(&buf[0], e.size);
  if (f.gcount() != e.size) throw std::runtime_error("tensor bundle: short read for " + name);
  if (crc32c_mask(crc32c(buf.data(), buf.size())) != e.crc)
    throw std::runtime_error("tensor bundle: checksum mismatch for " + name);
  return buf;
}

// ------------------------------------------------------------------------------------------
void write_checkpoint_state(const std::string& dir, const std::string& latest, const std::vector<std::string>& all) {
  std::string s = "model_checkpoint_path: \"" + latest + "\"\n";
  for (const auto& p : all) s += "all_model_checkpoint_paths: \"" + p + "\"\n";
  write_file_atomic(dir + "/checkpoint", s);
}

bool read_checkpoint_state(const std::string& dir, std::string* latest, std::vector<std::string>* all) {
  std::ifstream f(dir + "/checkpoint");
  if (!f) return false;
  std::string line;
  bool ok = false;
  while (std::getline(f, line)) {
    const size_t q0 = line.find('"'), q1 = line.rfind('"');
    if (q0 == std::string::npos || q1 <= q0) continue;
    const std::string val = line.substr(q0 + 1, q1 - q0 - 1);
    if (line.compare(0, 22, "model_checkpoint_path:") == 0) {
      *latest = val;
      ok = true;
    } else if (line.compare(0, 27, "all_model_checkpoint_paths:") == 0 && all) {
      all->push_back(val);
    }
  }
  return ok;
}

}  // namespace ckpt
}  // namespace dtf
