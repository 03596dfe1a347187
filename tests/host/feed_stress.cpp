// Host stress test of csrc/feed.h -- the queue between the encoder passes (one producer) and the row groups (consumers)
// of mt3_engine_transcribe -- with plain threads and no GPU.  The staging ring is modelled as an int per entry: the
// producer stamps an entry with the segment it encoded there when the chunk goes on offer; a consumer reads its entries
// at a RANDOM LATER TIME before it gives them back (as the refill copies run on the device until the next poll's event),
// so an entry overwritten before its release shows up as a wrong stamp.  Checked: every segment handed out exactly once,
// to one consumer, with the right entry; no overwrite before release; padding entries never on offer; termination with
// every consumer seeing `dry`; the failure path wakes everybody.
// Built with g++ by tests/test_feed_protocol.py.  Test infrastructure only.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <random>
#include <thread>
#include <vector>

#include "feed.h"

using namespace mt3feed;

// returns 0 on success, a positive error code otherwise
// prod_us / cons_us: upper bounds of the random delays of an encoder pass / between two polls; max_want: most finished slots a
// poll reports (0: cap + 2).  Slow consumers fill the ring, so that the producer has to WAIT for chunks to be given back.
extern "C" int feed_stress(int n_total, int slots, int cap, int min_batch, int consumers, unsigned seed, int fail_after_chunks,
                           int prod_us, int cons_us, int max_want) {
  Feed f;
  f.n_total = n_total;
  f.next_seg = slots < n_total ? slots : n_total;
  f.finished = n_total <= slots;
  std::vector<int> ring(static_cast<size_t>(kStageChunks) * cap, -1);     // entry -> segment whose data it holds
  std::vector<std::atomic<int>> taken_by(n_total);
  for (auto& t : taken_by) t.store(-1);
  std::atomic<int> err{0};
  auto bad = [&](int code) { int z = 0; err.compare_exchange_strong(z, code); };

  std::thread producer([&] {
    std::mt19937 rng(seed);
    int chunks = 0;
    bool failed = false;
    for (int q = 0;; ++q) {
      int first, n, pad;
      if (!feed_claim(f, q, cap, min_batch, &first, &n, &pad)) break;
      if (pad + n > cap && n >= min_batch) bad(1);
      // the "encoder pass": overwrite the chunk's entries (it must have been given back completely by now)
      std::this_thread::sleep_for(std::chrono::microseconds(rng() % (prod_us + 1)));
      const size_t base = static_cast<size_t>(q % kStageChunks) * cap;
      for (int i = 0; i < pad + n && i < cap; ++i) ring[base + i] = first - pad + i;
      if (fail_after_chunks > 0 && ++chunks == fail_after_chunks) {
        failed = true;
        break;
      }
      feed_publish(f, q, first, n, pad);
    }
    feed_finish(f, failed);
  });

  std::vector<std::thread> cons;
  std::atomic<int> dry_seen{0};
  for (int c = 0; c < consumers; ++c)
    cons.emplace_back([&, c] {
      std::mt19937 rng(seed * 31 + c);
      std::vector<FeedRange> held, got(kStageChunks + 2);
      for (;;) {
        // the previous poll's entries are read (the copy kernels) and only then given back
        std::this_thread::sleep_for(std::chrono::microseconds(rng() % (cons_us + 1)));
        for (const FeedRange& r : held)
          for (int i = 0; i < r.n; ++i)
            if (ring[static_cast<size_t>(r.seq % kStageChunks) * cap + r.entry0 + i] != r.first_seg + i) bad(2);   // overwritten early
        feed_release(f, held);
        held.clear();
        if (feed_failed(f)) return;
        bool dry = false;
        const int want = static_cast<int>(rng() % ((max_want > 0 ? max_want : cap + 2) + 1));   // finished slots at this poll (sometimes none)
        const int nr = feed_pop(f, want, got.data(), static_cast<int>(got.size()), &dry);
        int total = 0;
        for (int i = 0; i < nr; ++i) {
          const FeedRange& r = got[i];
          total += r.n;
          if (r.n <= 0 || r.entry0 < 0 || r.entry0 + r.n > r.batch || r.batch > cap) bad(3);
          for (int k = 0; k < r.n; ++k) {
            const int seg = r.first_seg + k;
            if (seg < slots || seg >= n_total) { bad(4); continue; }
            int none = -1;
            if (!taken_by[seg].compare_exchange_strong(none, c)) bad(5);   // handed out twice
          }
          held.push_back(r);
        }
        if (total > want) bad(6);
        if (dry) {
          ++dry_seen;
          // (held entries are given back at the consumer's exit, after its last "drain")
          for (const FeedRange& r : held)
            for (int i = 0; i < r.n; ++i)
              if (ring[static_cast<size_t>(r.seq % kStageChunks) * cap + r.entry0 + i] != r.first_seg + i) bad(2);
          feed_release(f, held);
          return;
        }
        if (nr == 0 && (rng() & 3) == 0) feed_wait(f);               // a group with nothing live sleeps until something is on offer
      }
    });
  producer.join();
  for (auto& t : cons) t.join();
  if (err.load()) return err.load();
  if (fail_after_chunks > 0) return f.failed ? 0 : 7;
  for (int s = slots; s < n_total; ++s)
    if (taken_by[s].load() < 0) return 8;                            // a segment nobody got
  if (dry_seen.load() != consumers) return 9;
  if (f.refills != (n_total > slots ? n_total - slots : 0)) return 10;
  return 0;
}
