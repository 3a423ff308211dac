// Self-test of the host C++ layer, meant to be built with sanitizers:
//   g++ -fsanitize=address,undefined ...   (memory errors, UB in the table/CRC/proto code)
//   g++ -fsanitize=thread ...              (data races in the shared-memory control board)
// It drives the checkpoint writer/reader through a round trip with many tensors and a corrupted
// file, and hammers the control board's arrival protocol, tokens, doorbells and barrier from
// several threads under uneven timing. Exit code 0 = all checks passed.
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../ckpt/tensor_bundle.h"
#include "../host/board.h"

#define CHECK(cond)                                                                  \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      std::fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                                  \
    }                                                                                \
  } while (0)

using namespace dtf;

static void test_bundle(const std::string& dir) {
  const std::string prefix = dir + "/model.ckpt-7";
  std::mt19937 rng(5);
  std::vector<std::pair<std::string, std::vector<float>>> tensors;
  {
    ckpt::BundleWriter w(prefix);
    for (int i = 0; i < 300; ++i) {
      std::string name = i == 0 ? "Variable" : "Variable_" + std::to_string(i);
      if (i % 3 == 0) name += "/Adam";
      std::vector<float> v(1 + rng() % 5000);
      for (auto& x : v) x = (float)(rng() % 1000) * 0.001f;
      w.add(name, ckpt::DT_FLOAT, {(int64_t)v.size()}, v.data(), v.size() * sizeof(float));
      tensors.emplace_back(name, std::move(v));
    }
    int32_t step = 7;
    w.add("global_step", ckpt::DT_INT32, {}, &step, sizeof(step));
    w.finish();
  }
  ckpt::BundleReader r(prefix);
  CHECK(r.entries().size() == 301);
  for (auto& [name, v] : tensors) {
    CHECK(r.contains(name));
    const std::string bytes = r.read(name);
    CHECK(bytes.size() == v.size() * sizeof(float));
    CHECK(std::memcmp(bytes.data(), v.data(), bytes.size()) == 0);
  }
  CHECK(r.entries().at("global_step").shape.empty());
  // table primitives: prefix compression across restart points and several blocks
  std::vector<std::pair<std::string, std::string>> kv;
  for (int i = 0; i < 2000; ++i) {
    char key[32];
    std::snprintf(key, sizeof(key), "key/%06d/suffix", i);
    kv.emplace_back(key, std::string(1 + i % 97, (char)('a' + i % 26)));
  }
  const std::string table = ckpt::build_table(kv, 4096, 16);
  CHECK(ckpt::parse_table(table) == kv);
  // a flipped byte in the data shard must be caught by the checksum
  {
    std::fstream f(prefix + ".data-00000-of-00001", std::ios::in | std::ios::out | std::ios::binary);
    f.seekp(10);
    char c;
    f.seekg(10);
    f.get(c);
    f.seekp(10);
    f.put((char)(c ^ 0x40));
  }
  bool threw = false;
  try {
    ckpt::BundleReader r2(prefix);
    (void)r2.read(tensors[0].first);
  } catch (const std::exception&) {
    threw = true;
  }
  CHECK(threw);
  ckpt::write_checkpoint_state(dir, "model.ckpt-7", {"model.ckpt-3", "model.ckpt-7"});
  std::string latest;
  std::vector<std::string> all;
  CHECK(ckpt::read_checkpoint_state(dir, &latest, &all));
  CHECK(latest == "model.ckpt-7" && all.size() == 2);
}

static void test_board(const std::string& name) {
  constexpr int kWorld = 6;
  constexpr int kSteps = 400;
  constexpr int kQuorum = 4;
  Board::unlink(name);
  Board creator(name, kWorld, true);
  std::atomic<int> in_mask_total{0};
  std::vector<std::thread> threads;
  for (int r = 0; r < kWorld; ++r) {
    threads.emplace_back([&, r] {
      Board b(name, kWorld, false);
      b.mark_ready(r);
      CHECK(b.wait_all_ready(30.0));
      std::mt19937 rng(100 + r);
      for (int step = 0; step < kSteps; ++step) {
        b.open_step(step);
        // uneven load: one rank is slow in a rotating pattern
        if ((step + r) % kWorld == 0) std::this_thread::sleep_for(std::chrono::microseconds(50 + rng() % 200));
        int order;
        if (b.stop_requested(step) && rng() % 2)
          order = b.arrive_aborted(step, r);
        else
          order = b.arrive(step, r, kQuorum);
        CHECK(order >= 0 && order < kWorld);
        CHECK(b.wait_arrivals(step, kWorld, 30.0));
        const uint64_t m = b.freeze_mask(step);
        CHECK(m == b.mask(step));
        CHECK(__builtin_popcountll(m) <= kQuorum);
        if (r == 0) in_mask_total += __builtin_popcountll(m);
        // tokens and doorbells in a ring
        b.put_tokens((r + 1) % kWorld, 1, step);
        int64_t tok_step = -1;
        CHECK(b.take_token(r, 30.0, &tok_step));
        b.heartbeat(r);
        CHECK(b.barrier(30.0));
      }
    });
  }
  for (auto& t : threads) t.join();
  CHECK(in_mask_total.load() > 0);
  Board::unlink(name);
}

int main() {
  char tmpl[] = "/tmp/dtf_selftest_XXXXXX";
  const char* dir = mkdtemp(tmpl);
  CHECK(dir != nullptr);
  test_bundle(dir);
  test_board("/dtf_selftest_" + std::to_string((long)getpid()));
  std::string cmd = std::string("rm -rf ") + dir;
  (void)!std::system(cmd.c_str());
  std::puts("host selftest OK");
  return 0;
}
