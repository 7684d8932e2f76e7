// Stub of the nanobind names that the reference's cmvm headers mention (bit_decompose.hh:5-15,
// mat_decompose.hh:11, mat_decompose.cc:139-169).  The Python-binding functions compiled against this stub
// are never called: oracle/ref_capi.cc talks to the reference's C++ API (api.hh) directly.
#pragma once
#include <cstddef>
#include <tuple>
#include <type_traits>
namespace nanobind {
struct numpy {};
namespace detail {
template <class... Ts> struct last_type;
template <class T> struct last_type<T> { using type = T; };
template <class T, class... Ts> struct last_type<T, Ts...> { using type = typename last_type<Ts...>::type; };
}  // namespace detail
template <class... Ts> class ndarray {
  public:
    using scalar = typename detail::last_type<Ts...>::type;
    ndarray() = default;
    template <class... A> ndarray(A &&...) {}
    std::size_t ndim() const { return 0; }
    std::size_t shape(std::size_t) const { return 0; }
    std::size_t size() const { return 0; }
    const scalar *data() const { return nullptr; }
};
struct capsule {
    template <class F> capsule(void *, F) {}
};
struct tuple {};
template <class... A> tuple make_tuple(A &&...) { return {}; }
namespace literals {}
}  // namespace nanobind
