// matrix.cpp — container part of the host mirror plus dot(), the seam where the
// reference hands its dense layers to OpenBLAS (reference src/matrix.cpp:106-122).
// Here dot() goes to the MI355X through the C ABI; nothing in this file computes
// a GEMM on the CPU.
#include <matrix.hpp>  // angle brackets: the -I order decides (reference headers in the drop-in build)

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "gnnvc.h"
#include "gnnvc_host.hpp"

matrix::matrix(size_t rows, size_t cols) : m(rows), n(cols), data(rows * cols) {}

size_t matrix::get_height() const { return m; }
size_t matrix::get_width() const { return n; }

void matrix::resize(size_t rows, size_t cols) {
    if (rows == m && cols == n) return;
    m = rows;
    n = cols;
    data.resize(rows * cols);
    selected_row.reset();
}

float &matrix::operator()(size_t i, size_t j) {
    assert(i < m && j < n);
    return data[i * n + j];
}
const float &matrix::operator()(size_t i, size_t j) const {
    assert(i < m && j < n);
    return data[i * n + j];
}

matrix &matrix::operator[](size_t i) {
    assert(i < m);
    selected_row = i;
    return *this;
}
const matrix &matrix::operator[](size_t i) const {
    assert(i < m);
    selected_row = i;
    return *this;
}
matrix &matrix::raw() {
    selected_row.reset();
    return *this;
}
const matrix &matrix::raw() const {
    selected_row.reset();
    return *this;
}

// cursor-dependent ranges: the selected row, or everything
std::vector<float>::iterator matrix::begin() { return data.begin() + (selected_row ? *selected_row * n : 0); }
std::vector<float>::iterator matrix::end() {
    return selected_row ? data.begin() + (*selected_row + 1) * n : data.end();
}
std::vector<float>::const_iterator matrix::begin() const {
    return data.cbegin() + (selected_row ? *selected_row * n : 0);
}
std::vector<float>::const_iterator matrix::end() const {
    return selected_row ? data.cbegin() + (*selected_row + 1) * n : data.cend();
}

std::vector<float>::iterator matrix::begin(size_t i) { return data.begin() + i * n; }
std::vector<float>::iterator matrix::end(size_t i) { return data.begin() + (i + 1) * n; }
std::vector<float>::const_iterator matrix::begin(size_t i) const { return data.cbegin() + i * n; }
std::vector<float>::const_iterator matrix::end(size_t i) const { return data.cbegin() + (i + 1) * n; }

std::ostream &operator<<(std::ostream &os, const matrix &mat) {
    os << mat.get_height() << " " << mat.get_width() << std::endl;
    for (size_t i = 0; i < mat.get_height(); ++i) {
        for (auto it = mat.begin(i); it != mat.end(i); ++it) os << *it << " ";
        os << std::endl;
    }
    return os;
}

std::istream &operator>>(std::istream &is, matrix &mat) {
    size_t h = 0, w = 0;
    is >> h >> w;
    mat.resize(h, w);
    for (auto it = mat.begin(0); it != mat.begin(0) + h * w; ++it) is >> *it;
    mat.raw();
    return is;
}

void dot(const matrix &A, const matrix &B, matrix &C, bool at, bool bt, float beta) {
    const size_t rows = at ? A.get_width() : A.get_height();
    const size_t inner = at ? A.get_height() : A.get_width();
    const size_t cols = bt ? B.get_height() : B.get_width();
    assert(inner == (bt ? B.get_width() : B.get_height()));
    C.resize(rows, cols);
    if (rows * cols == 0) return;
    const int rc = gnnvc_sgemm(gnnvc_host::ops_engine(), at, bt, (uint32_t)rows, (uint32_t)cols,
                               (uint32_t)inner, A.data.data(), (uint32_t)A.get_width(),
                               B.data.data(), (uint32_t)B.get_width(), beta, C.data.data(),
                               (uint32_t)C.get_width());
    gnnvc_host::check(rc, "dot/gnnvc_sgemm");
}
