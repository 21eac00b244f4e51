// Host-side batch sampler: draws one training batch's window descriptors from the SAME two
// Mersenne-Twister streams the reference consumes, in the same order, so that the
// SpecAugment mask indices and the chosen windows are bit-identical to
// reference microwakeword/data.py:531-597 under identical seeds:
//
//   1. random.choices(providers, sampling_weights, k=B)                    data.py:541-553
//   2. per slot: [random.choice(fixed_right_cutoffs)]                      data.py:252-253
//               random.choice(feature_sets["training"])                    data.py:255
//               [np.random.randint(0, L-T)]  (strategy "random", L > T)    data.py:100
//               per time mask: np.random.uniform(0,tmax) ; random.randint  data.py:61-64
//               per freq mask: np.random.uniform(0,fmax) ; random.randint  data.py:66-69
//   3. np.random.shuffle(indices)                                          data.py:591-595
//
// The Python host hands over the raw MT19937 states (random.getstate() / np.random.get_state()),
// this code advances them, and the host installs them back, so Python code that keeps using the
// global RNGs afterwards sees exactly the stream positions the reference would have left.
//
// Algorithms restated from CPython 3.10 Lib/random.py + Modules/_randommodule.c and numpy
// (legacy RandomState over MT19937: random_standard_uniform, masked-rejection bounded ints).
#include <stdint.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "../../include/mww.h"

namespace {

struct MT {
  uint32_t* s;   // 624 words
  uint32_t* pos; // index
  inline uint32_t next() {
    if (*pos >= 624) regen();
    uint32_t y = s[(*pos)++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  void regen() {
    const uint32_t N = 624, M = 397;
    uint32_t kk;
    for (kk = 0; kk < N - M; kk++) {
      uint32_t y = (s[kk] & 0x80000000u) | (s[kk + 1] & 0x7fffffffu);
      s[kk] = s[kk + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < N - 1; kk++) {
      uint32_t y = (s[kk] & 0x80000000u) | (s[kk + 1] & 0x7fffffffu);
      s[kk] = s[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    uint32_t y = (s[N - 1] & 0x80000000u) | (s[0] & 0x7fffffffu);
    s[N - 1] = s[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    *pos = 0;
  }
  // 53-bit double in [0,1): identical in CPython random.random() and numpy legacy random_sample()
  inline double real53() {
    uint32_t a = next() >> 5, b = next() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }
};

inline int bit_length(uint32_t n) {
  int k = 0;
  while (n) { ++k; n >>= 1; }
  return k;
}

// CPython Random._randbelow_with_getrandbits (n < 2^32 here)
inline uint32_t py_randbelow(MT& r, uint32_t n) {
  if (n == 0) return 0;
  const int k = bit_length(n);
  uint32_t v = r.next() >> (32 - k);
  while (v >= n) v = r.next() >> (32 - k);
  return v;
}

// numpy legacy bounded integer in [0, rng] (inclusive), masked rejection, one 32-bit draw per try
inline uint32_t np_interval(MT& r, uint32_t rng) {
  if (rng == 0) return 0;
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint32_t v;
  while ((v = (r.next() & mask)) > rng) {}
  return v;
}

}  // namespace

extern "C" int mww_sample_training_batch(const mww_sampler_desc* d, uint32_t* py_state, uint32_t* np_state, int B, int T,
                                         int tmax, int tcount, int fmax, int fcount, int32_t default_strategy,
                                         int32_t apply_order, mww_window* out_windows, int32_t* out_masks,
                                         int32_t* out_provider, int32_t* out_sample, int32_t* out_order) {
  if (!d || !py_state || !np_state || B < 0 || d->n_providers <= 0) return MWW_ERR_INVALID;
  MT py{py_state, py_state + 624}, np{np_state, np_state + 624};
  const int n = d->n_providers;
  const int nm = tcount + fcount;
  // 1. random.choices: bisect_right(cum_weights, random()*total, 0, n-1)
  double total = 0.0;
  double cum[64];
  if (n > 64) return MWW_ERR_INVALID;
  for (int i = 0; i < n; ++i) { total += d->sampling_weight[i]; cum[i] = total; }
  total = cum[n - 1] + 0.0;
  for (int j = 0; j < B; ++j) {
    const double x = py.real53() * total;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi) / 2;
      if (x < cum[mid]) hi = mid; else lo = mid + 1;
    }
    out_provider[j] = lo;
  }
  // 2. per-slot draws, in draw order (slot j) — the final permutation is applied afterwards
  for (int j = 0; j < B; ++j) {
    const int p = out_provider[j];
    const int strat = default_strategy >= 0 ? default_strategy : d->strategy[p];
    int cutoff = 0;
    if (strat == MWW_STRATEGY_FIXED_RIGHT_CUTOFF) {
      const int nc = d->cutoff_offsets[p + 1] - d->cutoff_offsets[p];
      cutoff = d->cutoffs[d->cutoff_offsets[p] + (int)py_randbelow(py, (uint32_t)nc)];
    }
    const int64_t s0 = d->set_offsets[p], s1 = d->set_offsets[p + 1];
    const int64_t pick = s0 + (int64_t)py_randbelow(py, (uint32_t)(s1 - s0));
    const int len = d->set_len[pick];
    int off = 0, copy = T, pad = 0;
    if (len > T) {
      switch (strat) {
        case MWW_STRATEGY_RANDOM: off = (int)np_interval(np, (uint32_t)(len - T - 1)); break;  // randint(0, L-T): high exclusive
        case MWW_STRATEGY_TRUNCATE_START: off = len - T; break;
        case MWW_STRATEGY_TRUNCATE_END: off = 0; break;
        case MWW_STRATEGY_FIXED_RIGHT_CUTOFF:
          off = len - T - cutoff;
          if (off < 0) return MWW_ERR_INVALID;  // the reference would produce a short window here
          break;
        default: return MWW_ERR_INVALID;        // "none" cannot form a fixed-length batch
      }
    } else {
      copy = len;
      pad = T - len;
    }
    mww_window w;
    w.store = d->set_store[pick];
    w.pad_rows = pad;
    w.copy_rows = copy;
    w.reserved = 0;
    w.src_elem = d->set_src_elem[pick] + (int64_t)off * 40;
    out_windows[j] = w;
    out_sample[j] = (int32_t)(pick - s0);
    int32_t* m = out_masks + (size_t)j * nm * 2;
    for (int i = 0; i < tcount; ++i) {
      const int t = (int)(0.0 + (double)tmax * np.real53());   // int(np.random.uniform(0, tmax))
      const int t0 = (int)py_randbelow(py, (uint32_t)(T - t + 1));  // random.randint(0, T - t)
      m[2 * i] = t0;
      m[2 * i + 1] = t;
    }
    for (int i = 0; i < fcount; ++i) {
      const int f = (int)(0.0 + (double)fmax * np.real53());
      const int f0 = (int)py_randbelow(py, (uint32_t)(40 - f + 1));
      m[2 * (tcount + i)] = f0;
      m[2 * (tcount + i) + 1] = f;
    }
  }
  // 3. np.random.shuffle(arange(B)): Fisher-Yates from the top with random_interval(i)
  for (int j = 0; j < B; ++j) out_order[j] = j;
  for (int i = B - 1; i >= 1; --i) {
    const int jx = (int)np_interval(np, (uint32_t)i);
    const int32_t tmp = out_order[i];
    out_order[i] = out_order[jx];
    out_order[jx] = tmp;
  }
  if (apply_order) {
    // hand the batch over in its final order: output slot j = draw out_order[j]
    std::vector<mww_window> w(out_windows, out_windows + B);
    std::vector<int32_t> m(out_masks, out_masks + (size_t)B * nm * 2), pr(out_provider, out_provider + B),
        sm(out_sample, out_sample + B);
    for (int j = 0; j < B; ++j) {
      const int src = out_order[j];
      out_windows[j] = w[src];
      out_provider[j] = pr[src];
      out_sample[j] = sm[src];
      for (int k = 0; k < nm * 2; ++k) out_masks[(size_t)j * nm * 2 + k] = m[(size_t)src * nm * 2 + k];
    }
  }
  return MWW_OK;
}

// exposes the two primitive streams so the tests can compare them with CPython / numpy directly
extern "C" int mww_rng_selftest(uint32_t* state, int which, int n, double* out_real, uint32_t* out_int, uint32_t bound) {
  MT r{state, state + 624};
  for (int i = 0; i < n; ++i) {
    if (which == 0) out_real[i] = r.real53();
    else if (which == 1) out_int[i] = py_randbelow(r, bound);
    else out_int[i] = np_interval(r, bound);
  }
  return MWW_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Batch prefetcher: the draws of get_data("training") for step n+1 .. n+depth are made by a worker thread while
// the launching thread enqueues step n (reference loop: microwakeword/train.py:276-299, where get_data and
// train_on_batch alternate on one thread).  The worker owns private copies of the two MT19937 streams and calls the
// same mww_sample_training_batch in the same order, so the sequence of batches is the one the synchronous path draws
// from those states (bit-exact: tests/engine_checks.py::check_prefetched_batches_match_synchronous_sampler).  The
// stream positions "after the last batch handed out" are kept with every slot, so the host can take the streams back
// at any batch boundary (policy change between training phases, checkpoint, switching the prefetch off).
#include <condition_variable>
#include <mutex>
#include <thread>

struct mww_prefetcher {
  // deep copies of the sampler description (the caller's arrays need not outlive the call)
  std::vector<double> sampling_weight;
  std::vector<int32_t> strategy, set_store, set_len, cutoff_offsets, cutoffs;
  std::vector<int64_t> set_offsets, set_src_elem;
  std::vector<float> label, weight;   // per provider (weight: penalty weight, or penalty x class weight when broadcast == 0)
  std::vector<float> class_weight;     // per provider (broadcast 1 / 2)
  int broadcast = 0;                   // how class and penalty weights combine: 0 per sample, 1 Keras' last-axis reading, 2 first-axis
  mww_sampler_desc d;
  int B = 0, T = 0, tmax = 0, tcount = 0, fmax = 0, fcount = 0, depth = 0;
  int32_t default_strategy = -1;
  uint32_t py[625], np_[625];         // worker's streams
  uint32_t last_py[625], last_np[625];   // positions after the last batch handed out
  struct Slot {
    std::vector<mww_window> win;
    std::vector<int32_t> masks, prov, samp, order;
    std::vector<float> y, w;
    uint32_t py[625], np_[625];
    int rc = MWW_OK;
  };
  std::vector<Slot> slots;
  int head = 0, tail = 0, ready = 0;   // ring: head = next slot handed out, tail = next slot filled
  bool held = false, stop = false;
  int64_t handed = 0;
  std::mutex mu;
  std::condition_variable cv_ready, cv_free;
  std::thread worker;

  void run() {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_free.wait(lk, [&] { return stop || ready < depth; });
        if (stop) return;
      }
      Slot& s = slots[tail];
      s.rc = mww_sample_training_batch(&d, py, np_, B, T, tmax, tcount, fmax, fcount, default_strategy, 1, s.win.data(),
                                       s.masks.data(), s.prov.data(), s.samp.data(), s.order.data());
      if (s.rc == MWW_OK) {
        // train.py:288-293 multiplies penalty[B] by class_weight(y)[B,1]: a [B,B] matrix W[i,j] = penalty_j cw(y_i) that Keras
        // reduces against the [B] losses.  broadcast 1: column means, w_j = penalty_j mean_i cw(y_i) (float64 like numpy);
        // 2: row means, w_i = cw(y_i) mean_j penalty_j; 0: the product per sample (already folded into `weight`)
        double mean_cw = 0.0, mean_pen = 0.0;
        if (broadcast != 0) {
          for (int j = 0; j < B; ++j) {
            mean_cw += (double)class_weight[s.prov[j]];
            mean_pen += (double)weight[s.prov[j]];
          }
          mean_cw /= B;
          mean_pen /= B;
        }
        for (int j = 0; j < B; ++j) {
          s.y[j] = label[s.prov[j]];
          if (broadcast == 1) s.w[j] = (float)((double)weight[s.prov[j]] * mean_cw);
          else if (broadcast == 2) s.w[j] = (float)((double)class_weight[s.prov[j]] * mean_pen);
          else s.w[j] = weight[s.prov[j]];
        }
      }
      memcpy(s.py, py, sizeof(py));
      memcpy(s.np_, np_, sizeof(np_));
      {
        std::lock_guard<std::mutex> lk(mu);
        tail = (tail + 1) % depth;
        ++ready;
      }
      cv_ready.notify_one();
      if (s.rc != MWW_OK) return;   // the error is handed out with this slot; nothing is drawn after it
    }
  }
};

extern "C" int mww_prefetch_create(const mww_sampler_desc* d, const float* provider_label, const float* provider_weight,
                                   const uint32_t* py_state, const uint32_t* np_state, int B, int T, int tmax, int tcount,
                                   int fmax, int fcount, int32_t default_strategy, int depth, mww_prefetcher** out) {
  return mww_prefetch_create_weighted(d, provider_label, provider_weight, nullptr, 0, py_state, np_state, B, T, tmax, tcount, fmax,
                                      fcount, default_strategy, depth, out);
}

extern "C" int mww_prefetch_create_weighted(const mww_sampler_desc* d, const float* provider_label, const float* provider_weight,
                                            const float* provider_class_weight, int broadcast, const uint32_t* py_state,
                                            const uint32_t* np_state, int B, int T, int tmax, int tcount, int fmax, int fcount,
                                            int32_t default_strategy, int depth, mww_prefetcher** out) {
  if (!d || !provider_label || !provider_weight || !py_state || !np_state || !out || B <= 0 || depth < 1 || depth > 16 ||
      d->n_providers <= 0 || d->n_providers > 64 || tcount < 0 || fcount < 0 || broadcast < 0 || broadcast > 2 ||
      (broadcast != 0 && !provider_class_weight))
    return MWW_ERR_INVALID;
  mww_prefetcher* p = new mww_prefetcher();
  const int n = d->n_providers;
  p->sampling_weight.assign(d->sampling_weight, d->sampling_weight + n);
  p->strategy.assign(d->strategy, d->strategy + n);
  p->set_offsets.assign(d->set_offsets, d->set_offsets + n + 1);
  const int64_t ns = p->set_offsets[n];
  p->set_store.assign(d->set_store, d->set_store + ns);
  p->set_src_elem.assign(d->set_src_elem, d->set_src_elem + ns);
  p->set_len.assign(d->set_len, d->set_len + ns);
  p->cutoff_offsets.assign(d->cutoff_offsets, d->cutoff_offsets + n + 1);
  const int nc = p->cutoff_offsets[n];
  p->cutoffs.assign(d->cutoffs, d->cutoffs + (nc > 0 ? nc : 0));   // (no provider with fixed_right_cutoffs: the array may be empty or null)
  if (p->cutoffs.empty()) p->cutoffs.push_back(0);                  // data() stays non-null
  p->label.assign(provider_label, provider_label + n);
  p->weight.assign(provider_weight, provider_weight + n);
  if (provider_class_weight) p->class_weight.assign(provider_class_weight, provider_class_weight + n);
  p->broadcast = broadcast;
  p->d.n_providers = n;
  p->d.sampling_weight = p->sampling_weight.data();
  p->d.strategy = p->strategy.data();
  p->d.set_offsets = p->set_offsets.data();
  p->d.set_store = p->set_store.data();
  p->d.set_src_elem = p->set_src_elem.data();
  p->d.set_len = p->set_len.data();
  p->d.cutoff_offsets = p->cutoff_offsets.data();
  p->d.cutoffs = p->cutoffs.data();
  p->B = B; p->T = T; p->tmax = tmax; p->tcount = tcount; p->fmax = fmax; p->fcount = fcount;
  p->default_strategy = default_strategy;
  p->depth = depth;
  memcpy(p->py, py_state, sizeof(p->py));
  memcpy(p->np_, np_state, sizeof(p->np_));
  memcpy(p->last_py, py_state, sizeof(p->py));
  memcpy(p->last_np, np_state, sizeof(p->np_));
  const int nm = tcount + fcount;
  p->slots.resize(depth);
  for (auto& s : p->slots) {
    s.win.resize(B);
    s.masks.resize((size_t)B * (nm > 0 ? nm : 1) * 2);
    s.prov.resize(B); s.samp.resize(B); s.order.resize(B);
    s.y.resize(B); s.w.resize(B);
  }
  p->worker = std::thread([p] { p->run(); });
  *out = p;
  return MWW_OK;
}

// blocks until the next batch is drawn; the pointers stay valid until mww_prefetch_release
extern "C" int mww_prefetch_acquire(mww_prefetcher* p, const mww_window** win, const int32_t** masks, const float** y,
                                    const float** w, const int32_t** provider, const int32_t** sample) {
  if (!p) return MWW_ERR_INVALID;
  std::unique_lock<std::mutex> lk(p->mu);
  if (p->held) return MWW_ERR_STATE;
  p->cv_ready.wait(lk, [&] { return p->ready > 0; });
  mww_prefetcher::Slot& s = p->slots[p->head];
  if (s.rc != MWW_OK) return s.rc;   // stays at the head: every later acquire reports it too
  p->held = true;
  p->handed += 1;
  memcpy(p->last_py, s.py, sizeof(s.py));
  memcpy(p->last_np, s.np_, sizeof(s.np_));
  if (win) *win = s.win.data();
  if (masks) *masks = s.masks.data();
  if (y) *y = s.y.data();
  if (w) *w = s.w.data();
  if (provider) *provider = s.prov.data();
  if (sample) *sample = s.samp.data();
  return MWW_OK;
}

extern "C" int mww_prefetch_release(mww_prefetcher* p) {
  if (!p) return MWW_ERR_INVALID;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->held) return MWW_ERR_STATE;
    p->held = false;
    p->head = (p->head + 1) % p->depth;
    --p->ready;
  }
  p->cv_free.notify_one();
  return MWW_OK;
}

// the two streams as they stood after the last batch handed out (batches drawn ahead of it are not counted)
extern "C" int64_t mww_prefetch_rng_state(mww_prefetcher* p, uint32_t* py_state, uint32_t* np_state) {
  if (!p) return MWW_ERR_INVALID;
  std::lock_guard<std::mutex> lk(p->mu);
  if (py_state) memcpy(py_state, p->last_py, sizeof(p->last_py));
  if (np_state) memcpy(np_state, p->last_np, sizeof(p->last_np));
  return p->handed;
}

extern "C" int mww_prefetch_shape(const mww_prefetcher* p, int* B, int* n_time_masks, int* n_freq_masks) {
  if (!p) return MWW_ERR_INVALID;
  if (B) *B = p->B;
  if (n_time_masks) *n_time_masks = p->tcount;
  if (n_freq_masks) *n_freq_masks = p->fcount;
  return MWW_OK;
}

extern "C" void mww_prefetch_destroy(mww_prefetcher* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->stop = true;
  }
  p->cv_free.notify_all();
  if (p->worker.joinable()) p->worker.join();
  delete p;
}
