// xt_shim.hpp -- a minimal, EAGER stand-in for the parts of xtensor 0.27 that the reference's
// CMVM sources use as an ndarray container (no CMVM arithmetic lives in xtensor).
//
// Purpose: let oracle/Makefile compile /root/reference/src/da4ml/_binary/cmvm/{api,cmvm_core,state_opr,
// indexers,bit_decompose,mat_decompose}.cc *as they lie* into oracle/_ref/libref.so, so the restated
// oracle and the HIP path can be pinned against the reference's own code.  Test infrastructure only.
//
// Semantics kept: row-major dense arrays, numpy broadcasting, C++ usual-arithmetic-conversion result
// types for element-wise operators (e.g. int8 - int8 -> int, float * double -> double), static_cast on
// assignment into a typed container, std:: overloads for pow/log2/ceil/abs.  Everything is evaluated
// immediately (xtensor is lazy; for the pure element-wise expressions used here the values are identical).
#pragma once

#include <algorithm>
#include <array>
#include <bit>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <functional>
#include <initializer_list>
#include <iostream>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace xt {

using shape_t = std::vector<std::size_t>;
using stride_t = std::vector<std::ptrdiff_t>;

template <class T> using store_t = std::conditional_t<std::is_same_v<T, bool>, unsigned char, T>;

inline std::size_t prod(const shape_t &s) {
    std::size_t n = 1;
    for (auto v : s) n *= v;
    return n;
}
inline stride_t dense_strides(const shape_t &s) {
    stride_t st(s.size());
    std::ptrdiff_t acc = 1;
    for (std::size_t i = s.size(); i-- > 0;) {
        st[i] = acc;
        acc *= (std::ptrdiff_t)s[i];
    }
    return st;
}

template <class T> class xview;

template <class T> class xarray {
  public:
    using value_type = T;
    using S = store_t<T>;

    xarray() = default;
    explicit xarray(const shape_t &shape) : shape_(shape), data_(prod(shape)) {}
    xarray(const shape_t &shape, T fill) : shape_(shape), data_(prod(shape), (S)fill) {}
    xarray(std::initializer_list<T> values) : shape_{values.size()}, data_(values.begin(), values.end()) {}
    template <class U> xarray(const xarray<U> &o) : shape_(o.shape()), data_(o.size()) {
        for (std::size_t i = 0; i < data_.size(); ++i) data_[i] = static_cast<S>(o.flat(i));
    }
    template <class U> xarray(const xview<U> &v);
    template <class U> xarray &operator=(const xarray<U> &o) {
        shape_ = o.shape();
        data_.resize(o.size());
        for (std::size_t i = 0; i < data_.size(); ++i) data_[i] = static_cast<S>(o.flat(i));
        return *this;
    }
    template <class U> xarray &operator=(const xview<U> &v) { return *this = xarray<U>(v); }

    const shape_t &shape() const { return shape_; }
    std::size_t shape(std::size_t i) const { return shape_[i]; }
    std::size_t dimension() const { return shape_.size(); }
    std::size_t size() const { return data_.size(); }
    S *data() { return data_.data(); }
    const S *data() const { return data_.data(); }
    auto begin() { return data_.begin(); }
    auto end() { return data_.end(); }
    auto begin() const { return data_.begin(); }
    auto end() const { return data_.end(); }
    S &flat(std::size_t i) { return data_[i]; }
    const S &flat(std::size_t i) const { return data_[i]; }
    void fill(T v) { std::fill(data_.begin(), data_.end(), (S)v); }

    void reshape(const shape_t &s) {
        if (prod(s) != data_.size()) throw std::runtime_error("xt_shim: reshape size mismatch");
        shape_ = s;
    }
    void reshape(std::initializer_list<std::size_t> s) { reshape(shape_t(s)); }

    template <class... I> S &operator()(I... idx) { return data_[offset(idx...)]; }
    template <class... I> const S &operator()(I... idx) const { return data_[offset(idx...)]; }

    template <class U> xarray &operator-=(const xarray<U> &o);

  private:
    template <class... I> std::size_t offset(I... idx) const {
        std::array<std::size_t, sizeof...(I)> ix{static_cast<std::size_t>(idx)...};
        if (sizeof...(I) != shape_.size()) throw std::runtime_error("xt_shim: index rank mismatch");
        std::size_t off = 0;
        for (std::size_t d = 0; d < ix.size(); ++d) off = off * shape_[d] + ix[d];
        return off;
    }
    shape_t shape_;
    std::vector<S> data_;
};

// ---------------------------------------------------------------- views
struct all_tag {};
struct newaxis_tag {};
struct range_tag {
    std::ptrdiff_t lo, hi;
};
inline all_tag all() { return {}; }
inline newaxis_tag newaxis() { return {}; }
template <class A, class B> range_tag range(A lo, B hi) { return {(std::ptrdiff_t)lo, (std::ptrdiff_t)hi}; }

template <class T> class xview {
  public:
    using value_type = T;
    using S = store_t<T>;
    S *base = nullptr;
    shape_t shape_;
    stride_t strides_;
    std::shared_ptr<xarray<T>> keep;  // owns the data when built from a temporary

    const shape_t &shape() const { return shape_; }
    std::size_t size() const { return prod(shape_); }
    std::size_t dimension() const { return shape_.size(); }

    template <class F> void for_each_offset(F &&f) const {
        std::size_t n = size(), nd = shape_.size();
        std::vector<std::size_t> idx(nd, 0);
        for (std::size_t k = 0; k < n; ++k) {
            std::ptrdiff_t off = 0;
            for (std::size_t d = 0; d < nd; ++d) off += (std::ptrdiff_t)idx[d] * strides_[d];
            f(k, off);
            for (std::size_t d = nd; d-- > 0;) {
                if (++idx[d] < shape_[d]) break;
                idx[d] = 0;
            }
        }
    }
    // assignment writes through to the viewed array (broadcasting the source)
    template <class U> xview &operator=(const xarray<U> &src);
    template <class U> xview &operator=(const xview<U> &src) { return *this = xarray<U>(src); }
    xview &operator=(const xview &src) { return *this = xarray<T>(src); }
    template <class U, std::enable_if_t<std::is_arithmetic_v<U>, int> = 0> xview &operator=(U v) {
        for_each_offset([&](std::size_t, std::ptrdiff_t off) { base[off] = static_cast<S>(v); });
        return *this;
    }
};

template <class T> template <class U> xarray<T>::xarray(const xview<U> &v) : shape_(v.shape()), data_(v.size()) {
    v.for_each_offset([&](std::size_t k, std::ptrdiff_t off) { data_[k] = static_cast<S>(v.base[off]); });
}

namespace detail {
template <class T> struct slicer {
    const shape_t &ashape;
    stride_t astr;
    std::size_t dim = 0;
    std::ptrdiff_t off = 0;
    shape_t oshape;
    stride_t ostr;
    void take(all_tag) {
        oshape.push_back(ashape[dim]);
        ostr.push_back(astr[dim]);
        ++dim;
    }
    void take(newaxis_tag) {
        oshape.push_back(1);
        ostr.push_back(0);
    }
    void take(range_tag r) {
        oshape.push_back((std::size_t)(r.hi - r.lo));
        ostr.push_back(astr[dim]);
        off += r.lo * astr[dim];
        ++dim;
    }
    template <class I, std::enable_if_t<std::is_integral_v<I>, int> = 0> void take(I i) {
        off += (std::ptrdiff_t)i * astr[dim];
        ++dim;
    }
    void finish() {
        while (dim < ashape.size()) take(all_tag{});
    }
};
}  // namespace detail

template <class T, class... Sl> xview<T> view(xarray<T> &a, Sl... sl) {
    detail::slicer<T> s{a.shape(), dense_strides(a.shape())};
    (s.take(sl), ...);
    s.finish();
    xview<T> v;
    v.base = a.data() + s.off;
    v.shape_ = s.oshape;
    v.strides_ = s.ostr;
    return v;
}
template <class T, class... Sl> xview<T> view(const xarray<T> &a, Sl... sl) {
    return view(const_cast<xarray<T> &>(a), sl...);  // read-only use in the reference
}
template <class T, class... Sl> xview<T> view(xarray<T> &&a, Sl... sl) {
    auto keep = std::make_shared<xarray<T>>(std::move(a));
    xview<T> v = view(*keep, sl...);
    v.keep = keep;
    return v;
}

// ---------------------------------------------------------------- traits / evaluation
template <class X> struct is_arr : std::false_type {};
template <class T> struct is_arr<xarray<T>> : std::true_type {};
template <class T> struct is_arr<xview<T>> : std::true_type {};
template <class X> inline constexpr bool is_arr_v = is_arr<std::decay_t<X>>::value;

template <class T> const xarray<T> &ev(const xarray<T> &a) { return a; }
template <class T> xarray<T> ev(const xview<T> &v) { return xarray<T>(v); }

inline shape_t bshape(const shape_t &a, const shape_t &b) {
    std::size_t n = std::max(a.size(), b.size());
    shape_t r(n);
    for (std::size_t i = 0; i < n; ++i) {
        std::size_t da = i < n - a.size() ? 1 : a[i - (n - a.size())];
        std::size_t db = i < n - b.size() ? 1 : b[i - (n - b.size())];
        if (da != db && da != 1 && db != 1) throw std::runtime_error("xt_shim: incompatible broadcast");
        r[i] = std::max(da, db);
    }
    return r;
}
inline stride_t bstrides(const shape_t &s, const shape_t &out) {
    stride_t d = dense_strides(s), r(out.size(), 0);
    std::size_t pad = out.size() - s.size();
    for (std::size_t i = 0; i < s.size(); ++i) r[i + pad] = s[i] == 1 ? 0 : d[i];
    return r;
}

template <class A, class B, class F> auto zip(const xarray<A> &a, const xarray<B> &b, F f) {
    using R = decltype(f(std::declval<A>(), std::declval<B>()));
    shape_t os = bshape(a.shape(), b.shape());
    xarray<R> out(os);
    if (a.shape() == b.shape()) {
        for (std::size_t i = 0; i < out.size(); ++i) out.flat(i) = f((A)a.flat(i), (B)b.flat(i));
        return out;
    }
    stride_t sa = bstrides(a.shape(), os), sb = bstrides(b.shape(), os);
    std::size_t nd = os.size(), n = out.size();
    std::vector<std::size_t> idx(nd, 0);
    for (std::size_t k = 0; k < n; ++k) {
        std::ptrdiff_t oa = 0, ob = 0;
        for (std::size_t d = 0; d < nd; ++d) {
            oa += (std::ptrdiff_t)idx[d] * sa[d];
            ob += (std::ptrdiff_t)idx[d] * sb[d];
        }
        out.flat(k) = f((A)a.flat(oa), (B)b.flat(ob));
        for (std::size_t d = nd; d-- > 0;) {
            if (++idx[d] < os[d]) break;
            idx[d] = 0;
        }
    }
    return out;
}
template <class A, class F> auto map(const xarray<A> &a, F f) {
    using R = decltype(f(std::declval<A>()));
    xarray<R> out(a.shape());
    for (std::size_t i = 0; i < out.size(); ++i) out.flat(i) = f((A)a.flat(i));
    return out;
}

template <class T> template <class U> xview<T> &xview<T>::operator=(const xarray<U> &src) {
    stride_t ss = bstrides(src.shape(), shape_);
    std::size_t nd = shape_.size(), n = size();
    std::vector<std::size_t> idx(nd, 0);
    for (std::size_t k = 0; k < n; ++k) {
        std::ptrdiff_t od = 0, os = 0;
        for (std::size_t d = 0; d < nd; ++d) {
            od += (std::ptrdiff_t)idx[d] * strides_[d];
            os += (std::ptrdiff_t)idx[d] * ss[d];
        }
        base[od] = static_cast<S>(src.flat(os));
        for (std::size_t d = nd; d-- > 0;) {
            if (++idx[d] < shape_[d]) break;
            idx[d] = 0;
        }
    }
    return *this;
}

// ---------------------------------------------------------------- element-wise operators
#define XT_SHIM_BINOP(OP)                                                                                     \
    template <class L, class R, std::enable_if_t<is_arr_v<L> && is_arr_v<R>, int> = 0>                        \
    auto operator OP(const L &l, const R &r) {                                                                \
        return zip(ev(l), ev(r), [](auto a, auto b) { return a OP b; });                                      \
    }                                                                                                         \
    template <class L, class R, std::enable_if_t<is_arr_v<L> && std::is_arithmetic_v<R>, int> = 0>            \
    auto operator OP(const L &l, R r) {                                                                       \
        return map(ev(l), [r](auto a) { return a OP r; });                                                    \
    }                                                                                                         \
    template <class L, class R, std::enable_if_t<std::is_arithmetic_v<L> && is_arr_v<R>, int> = 0>            \
    auto operator OP(L l, const R &r) {                                                                       \
        return map(ev(r), [l](auto b) { return l OP b; });                                                    \
    }
XT_SHIM_BINOP(+)
XT_SHIM_BINOP(-)
XT_SHIM_BINOP(*)
XT_SHIM_BINOP(<)
XT_SHIM_BINOP(>)
#undef XT_SHIM_BINOP

template <class A, std::enable_if_t<is_arr_v<A>, int> = 0> auto operator-(const A &a) {
    return map(ev(a), [](auto v) { return -v; });
}
template <class T> template <class U> xarray<T> &xarray<T>::operator-=(const xarray<U> &o) {
    if (o.shape() != shape_) throw std::runtime_error("xt_shim: -= shape mismatch");
    for (std::size_t i = 0; i < data_.size(); ++i) data_[i] = static_cast<S>(data_[i] - o.flat(i));
    return *this;
}

template <class To, class A, std::enable_if_t<is_arr_v<A>, int> = 0> auto cast(const A &a) {
    return map(ev(a), [](auto v) { return static_cast<To>(v); });
}
template <class A, std::enable_if_t<is_arr_v<A>, int> = 0> auto abs(const A &a) {
    return map(ev(a), [](auto v) { return std::abs(v); });
}
template <class A, std::enable_if_t<is_arr_v<A>, int> = 0> auto log2(const A &a) {
    return map(ev(a), [](auto v) { return std::log2(v); });
}
template <class A, std::enable_if_t<is_arr_v<A>, int> = 0> auto ceil(const A &a) {
    return map(ev(a), [](auto v) { return std::ceil(v); });
}
template <class B, class E, std::enable_if_t<std::is_arithmetic_v<B> && is_arr_v<E>, int> = 0>
auto pow(B base, const E &e) {
    return map(ev(e), [base](auto x) { return std::pow(base, x); });
}
template <class A, class B, std::enable_if_t<is_arr_v<A> && std::is_arithmetic_v<B>, int> = 0>
auto maximum(const A &a, B b) {
    return map(ev(a), [b](auto x) {
        using C = std::common_type_t<decltype(x), B>;
        return std::max<C>(x, b);
    });
}
template <class A, class B, std::enable_if_t<is_arr_v<A> && is_arr_v<B>, int> = 0> auto minimum(const A &a, const B &b) {
    return zip(ev(a), ev(b), [](auto x, auto y) {
        using C = std::common_type_t<decltype(x), decltype(y)>;
        return std::min<C>(x, y);
    });
}
template <class A, class B, std::enable_if_t<is_arr_v<A> && std::is_arithmetic_v<B>, int> = 0>
auto not_equal(const A &a, B b) {
    return map(ev(a), [b](auto x) { return x != b; });
}
template <class C, class A, class B> auto where(const C &c, const xarray<A> &a, const xarray<B> &b) {
    using R = std::common_type_t<A, B>;
    const auto &cc = ev(c);
    auto ab = zip(a, b, [](A x, B y) { return std::pair<R, R>(x, y); });  // broadcast the two branches together
    shape_t os = bshape(cc.shape(), ab.shape());
    xarray<R> out(os);
    stride_t sc = bstrides(cc.shape(), os), sab = bstrides(ab.shape(), os);
    std::size_t nd = os.size(), n = out.size();
    std::vector<std::size_t> idx(nd, 0);
    for (std::size_t k = 0; k < n; ++k) {
        std::ptrdiff_t oc = 0, oab = 0;
        for (std::size_t d = 0; d < nd; ++d) {
            oc += (std::ptrdiff_t)idx[d] * sc[d];
            oab += (std::ptrdiff_t)idx[d] * sab[d];
        }
        out.flat(k) = cc.flat(oc) ? ab.flat(oab).first : ab.flat(oab).second;
        for (std::size_t d = nd; d-- > 0;) {
            if (++idx[d] < os[d]) break;
            idx[d] = 0;
        }
    }
    return out;
}
template <class A, std::enable_if_t<is_arr_v<A>, int> = 0> bool any(const A &a) {
    const auto &x = ev(a);
    for (std::size_t i = 0; i < x.size(); ++i)
        if (x.flat(i)) return true;
    return false;
}

// ---------------------------------------------------------------- reductions
template <class T> struct scalar_result {
    T v;
    T operator()() const { return v; }
    operator T() const { return v; }
};
template <class A, std::enable_if_t<is_arr_v<A>, int> = 0> auto amax(const A &a) {
    const auto &x = ev(a);
    using T = typename std::decay_t<decltype(x)>::value_type;
    if (x.size() == 0) throw std::runtime_error("xt_shim: amax of empty array");
    T m = x.flat(0);
    for (std::size_t i = 1; i < x.size(); ++i) m = std::max<T>(m, x.flat(i));
    return scalar_result<T>{m};
}
template <class T, class Init, class F> xarray<T> reduce_axis(const xarray<T> &x, std::size_t axis, Init init, F f) {
    shape_t os;
    for (std::size_t d = 0; d < x.dimension(); ++d)
        if (d != axis) os.push_back(x.shape(d));
    std::size_t outer = 1, inner = 1, len = x.shape(axis);
    for (std::size_t d = 0; d < axis; ++d) outer *= x.shape(d);
    for (std::size_t d = axis + 1; d < x.dimension(); ++d) inner *= x.shape(d);
    xarray<T> out(os);
    for (std::size_t o = 0; o < outer; ++o)
        for (std::size_t i = 0; i < inner; ++i) {
            T acc = init(x, o * len * inner + i);
            for (std::size_t k = 1; k < len; ++k) acc = f(acc, (T)x.flat((o * len + k) * inner + i));
            out.flat(o * inner + i) = acc;
        }
    return out;
}
template <class A, std::enable_if_t<is_arr_v<A>, int> = 0> auto amin(const A &a, std::size_t axis) {
    const auto &x = ev(a);
    using T = typename std::decay_t<decltype(x)>::value_type;
    if (x.shape(axis) == 0) throw std::runtime_error("xt_shim: amin over empty axis");
    return reduce_axis<T>(
        x, axis, [](const xarray<T> &y, std::size_t i) { return (T)y.flat(i); }, [](T p, T q) { return std::min<T>(p, q); });
}
template <class A, std::enable_if_t<is_arr_v<A>, int> = 0> auto sum(const A &a, std::initializer_list<std::size_t> axes) {
    const auto &x = ev(a);
    using T = typename std::decay_t<decltype(x)>::value_type;
    if (axes.size() != 1) throw std::runtime_error("xt_shim: sum over one axis only");
    std::size_t axis = *axes.begin();
    if (x.shape(axis) == 0) {
        shape_t os;
        for (std::size_t d = 0; d < x.dimension(); ++d)
            if (d != axis) os.push_back(x.shape(d));
        return xarray<T>(os, T(0));
    }
    return reduce_axis<T>(
        x, axis, [](const xarray<T> &y, std::size_t i) { return (T)y.flat(i); }, [](T p, T q) { return (T)(p + q); });
}

// ---------------------------------------------------------------- generators / adaptors
template <class T> xarray<T> empty(const shape_t &s) { return xarray<T>(s); }
template <class T> xarray<T> zeros(const shape_t &s) { return xarray<T>(s, T(0)); }
template <class T> xarray<T> eye(std::size_t n) {
    xarray<T> r(shape_t{n, n}, T(0));
    for (std::size_t i = 0; i < n; ++i) r(i, i) = T(1);
    return r;
}
struct no_ownership {};
template <class P, class Sh> auto adapt(P *ptr, std::size_t size, no_ownership, const Sh &shape) {
    using T = std::remove_const_t<P>;
    xarray<T> r(shape_t(shape.begin(), shape.end()));
    if (r.size() != size) throw std::runtime_error("xt_shim: adapt size mismatch");
    std::copy(ptr, ptr + size, r.data());
    return r;
}
template <class F> auto vectorize(F f) {
    return [f](const auto &a) { return map(ev(a), [f](auto v) { return f(v); }); };
}

}  // namespace xt
