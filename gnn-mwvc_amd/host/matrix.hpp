// matrix.hpp — host-side dense fp32 matrix with the public interface of the
// reference's `class matrix` (reference include/matrix.hpp:6-43), so that code
// written against the reference (its driver, its layer structs) compiles
// unchanged against this header.  Row-major, element (i, j) at data[i*n + j].
//
// The one arithmetic entry point, dot(), is the reference's OpenBLAS seam
// (reference src/matrix.cpp:106-122); here it runs on the MI355X through the C
// ABI (include/gnnvc.h: gnnvc_sgemm) — see matrix.cpp.
#pragma once
#include <cstddef>
#include <iosfwd>
#include <optional>
#include <vector>

class matrix {
  public:
    using iterator = std::vector<float>::iterator;
    using const_iterator = std::vector<float>::const_iterator;

    matrix() = default;
    matrix(size_t m, size_t n);

    // shape
    size_t get_height() const;
    size_t get_width() const;
    void resize(size_t m, size_t n);  // keeps the contents when the shape is unchanged

    // element access
    float &operator()(size_t i, size_t j);
    const float &operator()(size_t i, size_t j) const;

    // Row cursor: M[i] selects row i for the argument-less begin()/end();
    // M.raw() clears the selection so they span the whole buffer.
    matrix &operator[](size_t i);
    const matrix &operator[](size_t i) const;
    matrix &raw();
    const matrix &raw() const;

    iterator begin();
    iterator end();
    const_iterator begin() const;
    const_iterator end() const;

    // explicit row ranges (do not touch the cursor)
    iterator begin(size_t i);
    iterator end(size_t i);
    const_iterator begin(size_t i) const;
    const_iterator end(size_t i) const;

    friend void dot(const matrix &A, const matrix &B, matrix &C, bool at, bool bt, float beta);

  private:
    // member names and order match the reference's so both headers describe one layout
    size_t m = 0, n = 0;
    mutable std::vector<float> data;
    mutable std::optional<size_t> selected_row = std::nullopt;
};

// "<h> <w>\n" then one line per row (reference src/matrix.cpp:87-104)
std::ostream &operator<<(std::ostream &os, const matrix &m);
std::istream &operator>>(std::istream &is, matrix &m);

// C = op(A) * op(B) + beta * C on the GPU (sequential-k fma chain per output).
void dot(const matrix &A, const matrix &B, matrix &C, bool at, bool bt, float beta);
