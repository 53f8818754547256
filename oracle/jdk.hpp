// ORACLE — TEST INFRASTRUCTURE ONLY. Nothing in the product path (wittgenstein_amd/, include/)
// may include, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker.
//
// JDK semantics the reference relies on but which are not under /root/reference (java.base,
// pinned by build.gradle:8-9 to Java 9). Restated from the published Javadoc contracts:
//   java.util.Random      (LCG, next(bits), nextInt, nextInt(bound) w/ rejection, nextBoolean,
//                          nextDouble, setSeed)  — call sites C/Network.java:32,55,377,430,
//                          C/Node.java:159,237,252, P/Handel.java:789
//   Collections.shuffle   — P/Handel.java:513,943  P/GSFSignature.java:470
//   java.util.BitSet      — everywhere in P/Handel.java, P/GSFSignature.java
// PARITY UNPINNED for seed-dependent outputs: no JVM in the build container, so the only anchors
// are the JDK known answers checked in tests/test_oracle_jdk.py.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>

namespace orc {

typedef int32_t jint;
typedef int64_t jlong;

struct IllegalArgumentException : std::runtime_error {
  explicit IllegalArgumentException(const std::string& s) : std::runtime_error(s) {}
};
struct IllegalStateException : std::runtime_error {
  explicit IllegalStateException(const std::string& s) : std::runtime_error(s) {}
};

// java.util.Random — 48-bit LCG per the class Javadoc.
class JRandom {
  static constexpr uint64_t MULT = 0x5DEECE66DULL;
  static constexpr uint64_t ADD = 0xBULL;
  static constexpr uint64_t MASK = (1ULL << 48) - 1;
  uint64_t seed_;

 public:
  uint64_t draws = 0;  // instrumentation only: number of next() calls
  explicit JRandom(jlong seed) { setSeed(seed); }
  void setSeed(jlong seed) { seed_ = ((uint64_t)seed ^ MULT) & MASK; }
  uint64_t rawState() const { return seed_; }
  void setRawState(uint64_t s) { seed_ = s & MASK; }

  jint next(int bits) {
    seed_ = (seed_ * MULT + ADD) & MASK;
    draws++;
    return (jint)(int64_t)(seed_ >> (48 - bits));  // (int)(seed >>> (48 - bits))
  }
  jint nextInt() { return next(32); }
  jint nextInt(jint bound) {
    if (bound <= 0) throw IllegalArgumentException("bound must be positive");
    jint r = next(31);
    jint m = bound - 1;
    if ((bound & m) == 0) {
      r = (jint)(((jlong)bound * (jlong)r) >> 31);
    } else {
      // for (int u = r; u - (r = u % bound) + m < 0; u = next(31));   (int overflow intended)
      jint u = r;
      for (;;) {
        r = u % bound;
        jint t = (jint)((uint32_t)u - (uint32_t)r + (uint32_t)m);
        if (t >= 0) break;
        u = next(31);
      }
    }
    return r;
  }
  bool nextBoolean() { return next(1) != 0; }
  double nextDouble() {
    jlong hi = (jlong)next(26);
    jlong lo = (jlong)next(27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
  }
};

// Collections.shuffle(list, rnd): for (i = size; i > 1; i--) swap(list, i-1, rnd.nextInt(i));
template <class T>
void jshuffle(std::vector<T>& l, JRandom& rd) {
  for (jint i = (jint)l.size(); i > 1; i--) {
    jint j = rd.nextInt(i);
    std::swap(l[i - 1], l[j]);
  }
}

// java.util.BitSet subset. Storage is a window [base_, base_+w_.size()) of 64-bit words (the
// JDK stores from word 0; the window is a memory optimisation with identical observable results).
class BitSet {
  std::vector<uint64_t> w_;
  int base_ = 0;  // index of first stored word

  uint64_t word(int wi) const {
    int k = wi - base_;
    return (k < 0 || k >= (int)w_.size()) ? 0ULL : w_[k];
  }
  void ensure(int lo, int hi) {  // make [lo, hi] addressable
    if (w_.empty()) {
      base_ = lo;
      w_.assign(hi - lo + 1, 0ULL);
      return;
    }
    if (lo < base_) {
      w_.insert(w_.begin(), base_ - lo, 0ULL);
      base_ = lo;
    }
    if (hi >= base_ + (int)w_.size()) w_.resize(hi - base_ + 1, 0ULL);
  }

 public:
  BitSet() {}
  bool get(int i) const { return (word(i >> 6) >> (i & 63)) & 1ULL; }
  void set(int i) {
    ensure(i >> 6, i >> 6);
    w_[(i >> 6) - base_] |= 1ULL << (i & 63);
  }
  void set(int i, bool v) {
    if (v)
      set(i);
    else
      clear(i);
  }
  void clear(int i) {
    int k = (i >> 6) - base_;
    if (k >= 0 && k < (int)w_.size()) w_[k] &= ~(1ULL << (i & 63));
  }
  void setRange(int from, int toExclusive) {  // BitSet.set(from, to)
    for (int i = from; i < toExclusive; i++) set(i);
  }
  void clear() { std::fill(w_.begin(), w_.end(), 0ULL); }
  int lo() const { return base_; }
  int hi() const { return base_ + (int)w_.size(); }
  void or_(const BitSet& o) {
    if (o.w_.empty()) return;
    int lo = -1, hi = -1;
    for (int k = 0; k < (int)o.w_.size(); k++)
      if (o.w_[k]) {
        if (lo < 0) lo = k;
        hi = k;
      }
    if (lo < 0) return;
    ensure(o.base_ + lo, o.base_ + hi);
    for (int k = lo; k <= hi; k++) w_[o.base_ + k - base_] |= o.w_[k];
  }
  void and_(const BitSet& o) {
    for (int k = 0; k < (int)w_.size(); k++) w_[k] &= o.word(base_ + k);
  }
  void andNot(const BitSet& o) {
    for (int k = 0; k < (int)w_.size(); k++) w_[k] &= ~o.word(base_ + k);
  }
  bool intersects(const BitSet& o) const {
    for (int k = 0; k < (int)w_.size(); k++)
      if (w_[k] & o.word(base_ + k)) return true;
    return false;
  }
  int cardinality() const {
    int c = 0;
    for (uint64_t x : w_) c += __builtin_popcountll(x);
    return c;
  }
  bool isEmpty() const {
    for (uint64_t x : w_)
      if (x) return false;
    return true;
  }
  bool equals(const BitSet& o) const {
    int a = std::min(base_, o.base_), b = std::max(hi(), o.hi());
    for (int k = a; k < b; k++)
      if (word(k) != o.word(k)) return false;
    return true;
  }
  int nextSetBit(int from) const {
    if (from < 0) from = 0;
    int wi = from >> 6;
    if (wi < base_) {
      wi = base_;
      from = wi * 64;
    }
    int top = hi();
    if (wi >= top) return -1;
    uint64_t x = word(wi) & (~0ULL << (from & 63));
    for (;;) {
      if (x) return wi * 64 + __builtin_ctzll(x);
      if (++wi >= top) return -1;
      x = word(wi);
    }
  }
  // raw word access for state dumps (word index in the JDK's from-zero numbering)
  uint64_t wordAt(int wi) const { return word(wi); }
  // BitSet.length(): the index of the highest set bit + 1
  int length() const {
    for (int k = (int)w_.size() - 1; k >= 0; k--)
      if (w_[k]) return (base_ + k) * 64 + 64 - __builtin_clzll(w_[k]);
    return 0;
  }
  // BitSet.hashCode(): h = 1234; for (i = wordsInUse; --i >= 0;) h ^= words[i] * (i + 1); return (int)((h >> 32) ^ h) — a zero
  // word contributes nothing, so the window and the JDK's wordsInUse do not matter
  jint hashCode() const {
    uint64_t h = 1234;
    for (int k = 0; k < (int)w_.size(); k++) h ^= w_[k] * (uint64_t)(base_ + k + 1);
    return (jint)(uint32_t)(((uint64_t)((int64_t)h >> 32)) ^ h);
  }
};

// java.util.HashSet<E> (a HashMap underneath, JDK 8+) where the ITERATION ORDER matters: bucket = (h ^ h >>> 16) & (cap - 1) of
// the hash the element had WHEN IT WAS ADDED (HashMap.Node.hash is final: an element mutated afterwards stays where it is and is
// found again only by its old hash), insertion order inside a bucket, tables of 16 doubling when the size passes 3/4 of the
// capacity (resize() splits every bucket into its low and high half, order preserved) or when a bucket would reach nine nodes in
// a table below 64 (treeifyBin resizes instead of treeifying there); clear() keeps the capacity. A bucket of nine in a table of
// 64 or more would become a tree (whose root moves to the bucket's front): not restated, an exception says so.
// E needs hashCode() and equals(const E&); elements are held by shared_ptr (Java references: identity first, as HashMap.putVal).
template <class E>
class JHashSet {
  struct Node {
    jint hash;
    std::shared_ptr<E> key;
  };
  std::vector<std::vector<Node>> tab_;
  int size_ = 0;
  static jint spread(jint h) { return h ^ (jint)((uint32_t)h >> 16); }
  void resize() {
    if (tab_.empty()) {
      tab_.assign(16, {});
      return;
    }
    const size_t oldCap = tab_.size();
    std::vector<std::vector<Node>> nt(oldCap * 2);
    for (size_t j = 0; j < oldCap; j++)
      for (Node& e : tab_[j]) nt[((uint32_t)e.hash & (uint32_t)oldCap) ? j + oldCap : j].push_back(std::move(e));
    tab_ = std::move(nt);
  }
  int threshold() const { return (int)(tab_.size() * 3 / 4); }

 public:
  int size() const { return size_; }
  bool isEmpty() const { return size_ == 0; }
  bool add(const std::shared_ptr<E>& k) {  // HashMap.putVal
    if (tab_.empty()) resize();
    const jint h = spread(k->hashCode());
    std::vector<Node>& b = tab_[(uint32_t)h & (uint32_t)(tab_.size() - 1)];
    for (const Node& e : b)
      if (e.hash == h && (e.key == k || k->equals(*e.key))) return false;
    const size_t before = b.size();
    b.push_back({h, k});
    if (before >= 8) {  // binCount >= TREEIFY_THRESHOLD - 1 in putVal's loop: treeifyBin
      if (tab_.size() >= 64) throw IllegalStateException("JHashSet: a bucket became a tree (not restated)");
      resize();
    }
    if (++size_ > threshold()) resize();
    return true;
  }
  bool remove(const std::shared_ptr<E>& k) {  // HashMap.removeNode: by the element's hash NOW
    if (tab_.empty()) return false;
    const jint h = spread(k->hashCode());
    std::vector<Node>& b = tab_[(uint32_t)h & (uint32_t)(tab_.size() - 1)];
    for (size_t i = 0; i < b.size(); i++)
      if (b[i].hash == h && (b[i].key == k || k->equals(*b[i].key))) {
        b.erase(b.begin() + i);
        size_--;
        return true;
      }
    return false;
  }
  void clear() {
    for (auto& b : tab_) b.clear();
    size_ = 0;
  }
  // the elements in iteration order (a snapshot: the callers here remove through removeAt / remove while walking a copy)
  std::vector<std::shared_ptr<E>> items() const {
    std::vector<std::shared_ptr<E>> out;
    for (const auto& b : tab_)
      for (const Node& e : b) out.push_back(e.key);
    return out;
  }
  // Iterator.remove() of the element the iterator just returned: unlinks THAT node (by identity, whatever its hash is now)
  void removeNode(const std::shared_ptr<E>& k) {
    for (auto& b : tab_)
      for (size_t i = 0; i < b.size(); i++)
        if (b[i].key == k) {
          b.erase(b.begin() + i);
          size_--;
          return;
        }
  }
  int capacity() const { return (int)tab_.size(); }
};

// BitSetUtils.include (C/utils/BitSetUtils.java:8-13)
inline bool bitsetInclude(const BitSet& big, const BitSet& small) {
  BitSet b = small;
  b.or_(big);
  return b.equals(big);
}

// MoreMath (C/utils/MoreMath.java:5-19)
inline int log2i(int n) {
  if (n <= 0) throw IllegalArgumentException("n=" + std::to_string(n));
  return 31 - __builtin_clz((unsigned)n);
}
inline int roundPow2(int n) {
  int res = 1 << (31 - __builtin_clz((unsigned)n));  // Integer.highestOneBit
  if (res != n) res <<= 1;
  return res;
}

}  // namespace orc
