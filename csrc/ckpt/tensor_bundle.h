// TensorFlow "tensor bundle" (checkpoint format V2) writer and reader, implemented from the
// on-disk format: `<prefix>.index` is a LevelDB-format table (prefix-compressed keys, restart
// points, per-block masked CRC32C trailers, 48-byte footer with magic 0xdb4775248b80fb57) mapping
// "" -> BundleHeaderProto and each variable name -> BundleEntryProto; `<prefix>.data-00000-of-00001`
// holds the raw little-endian tensor bytes back to back.
//
// This replaces tf.train.Saver's SaveV2/RestoreV2 kernels that the reference calls through
// (reference: DistributedLeNet/src/distributed_train.py:212,370-373; nn_eval.py:70-84).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace dtf {
namespace ckpt {

// TensorFlow DataType enum values for the dtypes a checkpoint of this framework contains.
enum DataType : int { DT_FLOAT = 1, DT_DOUBLE = 2, DT_INT32 = 3, DT_INT64 = 9 };

uint32_t crc32c(const void* data, size_t n, uint32_t init = 0);
uint32_t crc32c_mask(uint32_t crc);
uint32_t crc32c_unmask(uint32_t masked);

struct TensorEntry {
  int dtype = DT_FLOAT;
  std::vector<int64_t> shape;
  int64_t offset = 0;   // into the data shard
  int64_t size = 0;     // bytes
  uint32_t crc = 0;     // masked crc32c of the tensor bytes
};

class BundleWriter {
 public:
  explicit BundleWriter(const std::string& prefix);
  // Adds one tensor (raw little-endian bytes). Names may be added in any order; the index is
  // written sorted by name, the data file in insertion order (like TensorFlow's BundleWriter).
  void add(const std::string& name, int dtype, const std::vector<int64_t>& shape, const void* data, size_t bytes);
  // Writes `<prefix>.data-00000-of-00001` and `<prefix>.index` atomically (temp file + rename).
  void finish();

 private:
  std::string prefix_;
  std::string data_;
  std::map<std::string, TensorEntry> entries_;
  bool finished_ = false;
};

class BundleReader {
 public:
  explicit BundleReader(const std::string& prefix);
  const std::map<std::string, TensorEntry>& entries() const { return entries_; }
  bool contains(const std::string& name) const { return entries_.count(name) != 0; }
  // Reads the tensor bytes and verifies the stored checksum.
  std::string read(const std::string& name) const;
  int num_shards() const { return num_shards_; }

 private:
  std::string prefix_;
  std::map<std::string, TensorEntry> entries_;
  int num_shards_ = 1;
};

// LevelDB table primitives, exposed for golden-byte tests.
std::string build_table(const std::vector<std::pair<std::string, std::string>>& sorted_kv, size_t block_size = 262144,
                        int restart_interval = 16);
std::vector<std::pair<std::string, std::string>> parse_table(const std::string& file_bytes);

// `checkpoint` state file (text-format CheckpointState proto).
void write_checkpoint_state(const std::string& dir, const std::string& latest, const std::vector<std::string>& all);
bool read_checkpoint_state(const std::string& dir, std::string* latest, std::vector<std::string>* all);

}  // namespace ckpt
}  // namespace dtf
