// Python bindings of the host-only native code: TensorFlow tensor-bundle checkpoint I/O and the
// shared-memory control board. Builds with plain g++ and loads without a GPU.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "ckpt/tensor_bundle.h"
#include "host/board.h"

namespace py = pybind11;
using namespace dtf;

PYBIND11_MODULE(_host, m) {
  m.doc() = "host-side native runtime: TF tensor-bundle checkpoints, shared-memory control board";

  // ------------------------------------------------------------------ checkpoint
  m.def("crc32c", [](py::bytes b) {
    std::string s = b;
    return ckpt::crc32c(s.data(), s.size());
  });
  m.def("crc32c_mask", &ckpt::crc32c_mask);
  m.def("crc32c_unmask", &ckpt::crc32c_unmask);
  m.def("build_table", [](const std::vector<std::pair<py::bytes, py::bytes>>& kv, size_t block_size, int restart) {
    std::vector<std::pair<std::string, std::string>> v;
    for (auto& e : kv) v.emplace_back(std::string(e.first), std::string(e.second));
    return py::bytes(ckpt::build_table(v, block_size, restart));
  }, py::arg("kv"), py::arg("block_size") = 262144, py::arg("restart_interval") = 16);
  m.def("parse_table", [](py::bytes b) {
    std::vector<std::pair<py::bytes, py::bytes>> out;
    for (auto& e : ckpt::parse_table(std::string(b))) out.emplace_back(py::bytes(e.first), py::bytes(e.second));
    return out;
  });
  m.def("write_checkpoint_state", &ckpt::write_checkpoint_state);
  m.def("read_checkpoint_state", [](const std::string& dir) -> py::object {
    std::string latest;
    std::vector<std::string> all;
    if (!ckpt::read_checkpoint_state(dir, &latest, &all)) return py::none();
    return py::make_tuple(latest, all);
  });

  py::class_<ckpt::BundleWriter>(m, "BundleWriter")
      .def(py::init<const std::string&>())
      .def("add", [](ckpt::BundleWriter& w, const std::string& name, int dtype, const std::vector<int64_t>& shape,
                     py::buffer data) {
        py::buffer_info info = data.request();
        size_t bytes = size_t(info.itemsize);
        for (auto d : info.shape) bytes *= size_t(d);
        // require a C-contiguous buffer
        ssize_t expect = info.itemsize;
        for (int i = int(info.ndim) - 1; i >= 0; --i) {
          if (info.shape[i] > 1 && info.strides[i] != expect) throw std::runtime_error("BundleWriter.add: buffer must be contiguous");
          expect *= info.shape[i];
        }
        w.add(name, dtype, shape, info.ptr, bytes);
      })
      .def("finish", [](ckpt::BundleWriter& w) {
        py::gil_scoped_release rel;
        w.finish();
      });

  py::class_<ckpt::BundleReader>(m, "BundleReader")
      .def(py::init<const std::string&>())
      .def("names", [](const ckpt::BundleReader& r) {
        std::vector<std::string> v;
        for (auto& e : r.entries()) v.push_back(e.first);
        return v;
      })
      .def("contains", &ckpt::BundleReader::contains)
      .def("entry", [](const ckpt::BundleReader& r, const std::string& name) {
        auto it = r.entries().find(name);
        if (it == r.entries().end()) throw std::runtime_error("no tensor named " + name);
        const auto& e = it->second;
        return py::make_tuple(e.dtype, e.shape, e.offset, e.size, e.crc);
      })
      .def("read", [](const ckpt::BundleReader& r, const std::string& name) { return py::bytes(r.read(name)); });

  m.attr("DT_FLOAT") = int(ckpt::DT_FLOAT);
  m.attr("DT_DOUBLE") = int(ckpt::DT_DOUBLE);
  m.attr("DT_INT32") = int(ckpt::DT_INT32);
  m.attr("DT_INT64") = int(ckpt::DT_INT64);

  // ------------------------------------------------------------------ control board
  py::class_<Board>(m, "Board")
      .def(py::init<const std::string&, int, bool, uint64_t>(), py::arg("name"), py::arg("world"), py::arg("create"),
           py::arg("nonce") = 0)
      .def_static("unlink", &Board::unlink)
      .def_static("now_ns", &Board::now_ns)
      .def_property_readonly("world", &Board::world)
      .def_property_readonly("name", &Board::name)
      .def("mark_ready", &Board::mark_ready)
      .def("all_ready", &Board::all_ready)
      .def("wait_all_ready", [](Board& b, double t) { py::gil_scoped_release rel; return b.wait_all_ready(t); })
      .def("barrier", [](Board& b, double t) { py::gil_scoped_release rel; return b.barrier(t); }, py::arg("timeout_s") = 600.0)
      .def("global_step", &Board::global_step)
      .def("set_global_step", &Board::set_global_step)
      .def("advance_global_step", &Board::advance_global_step)
      .def("request_stop", &Board::request_stop)
      .def("should_stop", &Board::should_stop)
      .def("report_failure", &Board::report_failure)
      .def("failed_rank", &Board::failed_rank)
      .def("open_step", &Board::open_step)
      .def("arrive", &Board::arrive, py::arg("step"), py::arg("rank"), py::arg("quorum") = 0)
      .def("arrive_aborted", &Board::arrive_aborted)
      .def("arrivals", &Board::arrivals)
      .def("stop_requested", &Board::stop_requested)
      .def("raise_stop", &Board::raise_stop)
      .def("step_start_ns", &Board::step_start_ns)
      .def("freeze_mask", &Board::freeze_mask)
      .def("mask", &Board::mask)
      .def("wait_arrivals", [](Board& b, int64_t step, int count, double t) {
        py::gil_scoped_release rel;
        return b.wait_arrivals(step, count, t);
      })
      .def("put_tokens", &Board::put_tokens)
      .def("take_token", [](Board& b, int rank, double t) -> py::object {
        int64_t step = -1;
        bool ok;
        {
          py::gil_scoped_release rel;
          ok = b.take_token(rank, t, &step);
        }
        if (!ok) return py::none();
        return py::int_(step);
      })
      .def("token_depth", &Board::token_depth)
      .def("set_local_step", &Board::set_local_step)
      .def("local_step", &Board::local_step)
      .def("heartbeat", &Board::heartbeat)
      .def("last_heartbeat_ns", &Board::last_heartbeat_ns)
      .def("depart", &Board::depart)
      .def_property_readonly_static("DEPARTED", [](py::object) { return Board::kDeparted; })
      .def("shard_version", &Board::shard_version)
      .def("bump_shard_version", &Board::bump_shard_version)
      .def("set_shard_version", &Board::set_shard_version)
      .def("ring_doorbell", &Board::ring_doorbell)
      .def("poll_doorbell", &Board::poll_doorbell)
      .def("clear_doorbell", &Board::clear_doorbell)
      .def("request_save", &Board::request_save)
      .def("save_step", &Board::save_step)
      .def("agg_reset", &Board::agg_reset)
      .def("agg_arrive", [](Board& b, int64_t e, int rank, int add, int quorum, int abort_at, float scale, bool any_update) {
        Board::AggArrival r;
        {
          py::gil_scoped_release rel;
          r = b.agg_arrive(e, rank, add, quorum, abort_at, scale, any_update);
        }
        return py::make_tuple(r.status, r.order, r.closed != 0, r.open_u);
      }, py::arg("e"), py::arg("rank"), py::arg("add") = 1, py::arg("quorum") = 0, py::arg("abort_at") = 0,
         py::arg("scale") = 0.f, py::arg("any_update") = false)
      .def("agg_take_back", &Board::agg_take_back)
      .def("agg_close", &Board::agg_close)
      .def("agg_open_update", &Board::agg_open_update)
      .def("agg_open_count", &Board::agg_open_count)
      .def("agg_open_mask", &Board::agg_open_mask)
      .def("agg_entry", [](const Board& b, int64_t u) -> py::object {
        int count = 0;
        uint32_t mask = 0, owners = 0;
        float scale = 0.f;
        bool rec = false;
        if (!b.agg_entry(u, &count, &mask, &owners, &scale, &rec)) {
          if (rec) throw std::runtime_error("update log: update " + std::to_string(u) + " was recycled (replica lags more than the ring)");
          return py::none();
        }
        return py::make_tuple(count, mask, owners, scale);
      })
      .def("agg_wait_closed", [](const Board& b, int64_t u, double t) { py::gil_scoped_release rel; return b.agg_wait_closed(u, t); })
      .def("agg_publish_ready", &Board::agg_publish_ready)
      .def("agg_is_ready", &Board::agg_is_ready)
      .def("agg_wait_ready", [](const Board& b, int64_t u, uint32_t owners, double t) {
        py::gil_scoped_release rel;
        return b.agg_wait_ready(u, owners, t);
      })
      .def("agg_set_applied", &Board::agg_set_applied)
      .def("agg_applied", &Board::agg_applied)
      .def("agg_min_applied", &Board::agg_min_applied)
      .def("agg_wait_min_applied", [](const Board& b, int64_t v, double t) { py::gil_scoped_release rel; return b.agg_wait_min_applied(v, t); })
      .def("agg_abort", &Board::agg_abort)
      .def("agg_raise_abort", &Board::agg_raise_abort)
      .def("agg_mark_dead", &Board::agg_mark_dead)
      .def("agg_dead", &Board::agg_dead)
      .def("post_times", &Board::post_times)
      .def("get_times", [](const Board& b, int rank, int64_t iter) -> py::object {
        int64_t a = 0, c = 0;
        if (!b.get_times(rank, iter, &a, &c)) return py::none();
        return py::make_tuple(a, c);
      });
  m.attr("BOARD_MAX_RANKS") = kBoardMaxRanks;
  m.attr("BOARD_RING") = kBoardRing;
  m.attr("AGG_RING") = kAggRing;
  m.attr("AGG_MAX_RANKS") = kAggMaxRanks;
}
