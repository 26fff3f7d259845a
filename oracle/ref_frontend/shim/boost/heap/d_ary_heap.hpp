// TEST INFRASTRUCTURE (oracle/ref_frontend): the part of boost::heap::d_ary_heap<T, mutable_<true>, arity<D>, compare<Cmp>> that
// jps3d's graph_search.{h,cpp} uses (push -> handle, top, pop, empty, clear, increase(handle)), so that graph_search.cpp compiles
// untouched although Boost is absent from this image.
//
// The heap discipline is the one Boost.Heap documents and implements for d_ary_heap (boost/heap/d_ary_heap.hpp): an implicit
// D-ary tree in a vector; `compare(a, b)` true means a has LOWER priority than b; push = append + sift-up while
// compare(parent, child); pop = move the last element to the root + sift-down towards the highest-priority child (the first one
// among equals, as std::max_element finds it) while !compare(child, node); increase(handle) = sift-up from the element's place.
// jps3d's comparator is not a strict weak order on ties (|f1 - f2| <= 1e-6: smaller g first), so WHICH of several equal-cost
// paths is returned depends on this discipline; it could not be checked against a Boost build here.  Path cost does not.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdio>  // (Boost.Heap brings <cstdio> in transitively; graph_search.cpp relies on it for printf)
#include <vector>

namespace boost {
namespace heap {
template <bool B>
struct mutable_ {};
template <unsigned D>
struct arity { static constexpr unsigned value = D; };
template <class C>
struct compare { typedef C type; };

template <class T, class Mutable, class Arity, class Compare>
class d_ary_heap {
  struct Node {
    T value;
    std::size_t pos;
  };
  static constexpr std::size_t D = Arity::value;

public:
  class handle_type {
    friend class d_ary_heap;
    Node* n_ = nullptr;

  public:
    handle_type() {}
    T& operator*() const { return n_->value; }
  };
  d_ary_heap() {}
  d_ary_heap(const d_ary_heap&) = delete;
  d_ary_heap& operator=(const d_ary_heap&) = delete;
  ~d_ary_heap() { clear(); }

  bool empty() const { return q_.empty(); }
  std::size_t size() const { return q_.size(); }
  void clear() {
    for (Node* n : q_) delete n;
    q_.clear();
  }
  const T& top() const { return q_.front()->value; }
  handle_type push(const T& v) {
    Node* n = new Node{v, q_.size()};
    q_.push_back(n);
    siftup(q_.size() - 1);
    handle_type h;
    h.n_ = n;
    return h;
  }
  void pop() {
    Node* out = q_.front();
    std::swap(q_.front(), q_.back());
    q_.front()->pos = 0;
    q_.pop_back();
    delete out;
    if (!q_.empty()) siftdown(0);
  }
  void increase(handle_type h) { siftup(h.n_->pos); }
  void decrease(handle_type h) { siftdown(h.n_->pos); }
  void update(handle_type h) {
    const std::size_t i = h.n_->pos;
    if (i != 0 && cmp_(q_[(i - 1) / D]->value, q_[i]->value)) siftup(i);
    else siftdown(i);
  }

private:
  void swap_nodes(std::size_t a, std::size_t b) {
    std::swap(q_[a], q_[b]);
    q_[a]->pos = a;
    q_[b]->pos = b;
  }
  void siftup(std::size_t index) {
    while (index != 0) {
      const std::size_t parent = (index - 1) / D;
      if (cmp_(q_[parent]->value, q_[index]->value)) {
        swap_nodes(parent, index);
        index = parent;
      } else
        return;
    }
  }
  void siftdown(std::size_t index) {
    for (;;) {
      const std::size_t first = index * D + 1;
      if (first >= q_.size()) return;
      const std::size_t last = std::min(first + D - 1, q_.size() - 1);
      std::size_t best = first;  // std::max_element: the first of the highest-priority children
      for (std::size_t c = first + 1; c <= last; c++)
        if (cmp_(q_[best]->value, q_[c]->value)) best = c;
      if (!cmp_(q_[best]->value, q_[index]->value)) {
        swap_nodes(best, index);
        index = best;
      } else
        return;
    }
  }
  std::vector<Node*> q_;
  typename Compare::type cmp_;
};
}  // namespace heap
}  // namespace boost
