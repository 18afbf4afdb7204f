// sampling_module.cpp -- the reference's native plugin as a COMPILED extension module named `sampling`
// (sources/sampling.cpp:95-106 of the reference: PYBIND11_MODULE(sampling, m) with randint / seed /
// sample_negative / sample_negative_ByUser), so that code which loads the plugin by path the way
// utils.py:25-34 does -- import a module object called `sampling` and call those four names -- works unchanged.
//
// This file holds no arithmetic: every function forwards to the C ABI of liblgcn_hip.so
// (include/lgcn_hip.h, "Host: BPR triplet sampler"), which owns the glibc rand() stream.  The ctypes
// binding (sampling.py) and this module therefore share ONE generator state, and the GPU sampler's
// jump-ahead keeps both in step.  Built by build.py with g++ + pybind11 into sources/sampling<ext>.so.
//
// Differences from the reference's module, all additive:
//   * allPos may also be a (indptr int64, indices int32) CSR tuple: no per-user copy (the reference copies
//     every positive list into a vector<vector<int>> BY VALUE on each call, sampling.cpp:27);
//   * a user without positives raises ValueError instead of dying with SIGFPE on rand() % 0;
//   * import does not call srand(time(0)) (sampling.cpp:97): the stream starts where the library's is.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "lgcn_hip.h"

namespace py = pybind11;

namespace {

struct Csr {
    std::vector<int64_t> indptr_own; std::vector<int32_t> indices_own;     // filled when allPos was a list of lists
    py::array_t<int64_t, py::array::c_style | py::array::forcecast> indptr_arr;
    py::array_t<int32_t, py::array::c_style | py::array::forcecast> indices_arr;
    const int64_t *indptr = nullptr; const int32_t *indices = nullptr; int64_t rows = 0;
};

void to_csr(const py::object &allPos, Csr &c) {
    if (py::isinstance<py::tuple>(allPos) && py::len(allPos) == 2) {      // zero-copy CSR overload
        py::tuple t = allPos.cast<py::tuple>();
        c.indptr_arr = t[0].cast<py::array_t<int64_t, py::array::c_style | py::array::forcecast>>();
        c.indices_arr = t[1].cast<py::array_t<int32_t, py::array::c_style | py::array::forcecast>>();
        if (c.indptr_arr.ndim() != 1 || c.indices_arr.ndim() != 1 || c.indptr_arr.size() < 1)
            throw std::invalid_argument("allPos CSR: (indptr [rows+1], indices [nnz]) expected");
        c.indptr = c.indptr_arr.data(); c.indices = c.indices_arr.data(); c.rows = c.indptr_arr.size() - 1;
        if (c.indptr[0] != 0 || c.indptr[c.rows] != (int64_t)c.indices_arr.size())
            throw std::invalid_argument("allPos CSR: indptr does not match indices");
        return;
    }
    // the reference's form: a sequence of per-user id sequences (dataloader.py:178-180: CSR row slices)
    py::sequence seq = allPos.cast<py::sequence>();
    c.rows = (int64_t)py::len(seq);
    c.indptr_own.assign((size_t)c.rows + 1, 0);
    for (int64_t u = 0; u < c.rows; u++) {
        auto row = py::array_t<int32_t, py::array::c_style | py::array::forcecast>::ensure(seq[(size_t)u]);
        if (!row || row.ndim() > 1) throw std::invalid_argument("allPos: every entry must be a 1-d sequence of item ids");
        const int64_t n = row.size();
        c.indices_own.insert(c.indices_own.end(), row.data(), row.data() + n);
        c.indptr_own[(size_t)u + 1] = c.indptr_own[(size_t)u] + n;
    }
    if (c.indices_own.empty()) c.indices_own.push_back(0);     // keep the pointer valid for an empty matrix
    c.indptr = c.indptr_own.data(); c.indices = c.indices_own.data();
}

[[noreturn]] void fail(int rc, const char *what) {
    const char *msg = lgcn_last_error();
    std::string m = std::string(what) + " failed (rc=" + std::to_string(rc) + "): " + (msg ? msg : "");
    if (rc == 2 || rc == 3) throw py::value_error(m);
    throw std::runtime_error(m);
}

// sampling.cpp:27-56
py::array_t<int32_t> sample_negative(int user_num, int item_num, int64_t train_num, py::object allPos, int neg_num) {
    Csr c; to_csr(allPos, c);
    if (user_num <= 0 || c.rows < user_num) throw py::value_error("allPos has fewer rows than user_num");
    if (neg_num < 1) throw py::value_error("neg_num must be >= 1");
    const int64_t rows = (int64_t)user_num * (train_num / user_num);
    py::array_t<int32_t> S({(py::ssize_t)rows, (py::ssize_t)(2 + neg_num)});
    int rc;
    {
        py::gil_scoped_release nogil;        // (the reference holds the GIL throughout; nothing here touches Python objects)
        rc = lgcn_sample_negative(user_num, item_num, train_num, c.indptr, c.indices, neg_num, S.mutable_data());
    }
    if (rc) fail(rc, "sampling.sample_negative");
    return S;
}

// sampling.cpp:58-86
py::array_t<int32_t> sample_negative_ByUser(py::array_t<int32_t, py::array::c_style | py::array::forcecast> users,
                                            int item_num, py::object allPos, int neg_num) {
    Csr c; to_csr(allPos, c);
    if (users.ndim() != 1) throw py::value_error("users must be 1-d");
    const int64_t n = users.size();
    for (int64_t i = 0; i < n; i++)
        if (users.data()[i] < 0 || users.data()[i] >= c.rows) throw py::value_error("user id out of range");
    if (neg_num < 1) throw py::value_error("neg_num must be >= 1");
    py::array_t<int32_t> S({(py::ssize_t)n, (py::ssize_t)(2 + neg_num)});
    int rc;
    {
        py::gil_scoped_release nogil;
        rc = lgcn_sample_negative_by_user(users.data(), (int)n, item_num, c.indptr, c.indices, neg_num, S.mutable_data());
    }
    if (rc) fail(rc, "sampling.sample_negative_ByUser");
    return S;
}

}  // namespace

PYBIND11_MODULE(sampling, m) {
    m.doc() = "example plugin";      // the reference's docstring (sampling.cpp:98)
    m.def("randint", [](int end) { return lgcn_sampling_randint(end); }, "generate int between [0 end]", py::arg("end"));
    m.def("seed", [](unsigned int seed) { lgcn_sampling_seed(seed); }, "set random seed", py::arg("seed"));
    m.def("sample_negative", &sample_negative, "sampling negatives for all", py::arg("user_num"), py::arg("item_num"),
          py::arg("train_num"), py::arg("allPos"), py::arg("neg_num"));
    m.def("sample_negative_ByUser", &sample_negative_ByUser, "sampling negatives for given users", py::arg("users"),
          py::arg("item_num"), py::arg("allPos"), py::arg("neg_num"));
    m.attr("abi_version") = lgcn_abi_version();
}
