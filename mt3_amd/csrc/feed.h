// The segment queue of mt3_engine_transcribe (in-flight batching): plain host C++, no HIP -- engine.hip includes it, and
// tests/host/feed_stress.cpp drives it with one producer and four consumer threads on a box without a GPU.
//
//   producer  the CALLING thread, on the caller's stream: encoder passes over chunks of up to `cap` segments, each into one
//             chunk of the staging ring; a chunk is on offer once its pass has finished on the device
//   consumers the row groups' worker threads: at every poll a group takes as many encoded segments off the ring as it has
//             finished slots and issues the refill launches (decode_ops.hip) on its own stream; what it took at one
//             poll it gives back to the producer at the NEXT poll, when the event in between has proven the copies done
// Everything the two sides share is host state under one mutex -- no cross-stream events, nobody waits on the device
// for the other side.  The ring is sized so that a poll's demand is normally on offer (8 chunks of 64 segments).
#ifndef MT3_FEED_H_
#define MT3_FEED_H_

#include <condition_variable>
#include <mutex>
#include <vector>

namespace mt3feed {

constexpr int kStageChunks = 8;        // chunks of the staging ring

struct StageChunk {
  int first_seg = 0;   // first segment on offer in this chunk
  int n = 0;           // segments on offer
  int pad = 0;         // entries in front of them: a short last chunk is encoded together with the `pad` segments before
                       // it (already handed out earlier), so that every pass is one of >= min_batch segments and takes
                       // the same tiles as a full one -- a segment's numbers do not depend on where the corpus ends
  int batch = 0;       // encoder batch of the pass = pad + n = the plane stride of the chunk's [2][batch][H][T][64] blocks
  int taken = 0, released = 0;
};

struct FeedRange {
  int seq, first_seg, entry0, n, batch;
};

struct Feed {
  std::mutex mu;
  std::condition_variable cv;
  int n_total = 0;
  int next_seg = 0;          // first segment not yet handed to an encoder pass
  int produced = 0;          // chunks on offer so far: sequence numbers [0, produced); sequence q lives in chunk[q % kStageChunks]
  int head = 0;              // sequence number the consumers take from
  bool finished = false;     // the producer is done: nothing will be added
  bool failed = false;       // a group or the producer failed: everybody leaves
  StageChunk chunk[kStageChunks];
  int polls = 0, refills = 0, starved = 0;
};

// ---- consumer side
// up to `want` encoded segments off the ring (whole runs of one chunk each); *dry: nothing is left and nothing will come
inline int feed_pop(Feed& f, int want, FeedRange* out, int max_out, bool* dry) {
  std::lock_guard<std::mutex> lk(f.mu);
  int n_out = 0;
  ++f.polls;
  while (f.head < f.produced) {
    StageChunk& c = f.chunk[f.head % kStageChunks];
    if (c.taken == c.n) {
      ++f.head;
      continue;
    }
    if (want <= 0 || n_out >= max_out) break;
    const int avail = c.n - c.taken, take = avail < want ? avail : want;
    out[n_out++] = FeedRange{f.head, c.first_seg + c.taken, c.pad + c.taken, take, c.batch};
    c.taken += take;
    want -= take;
    f.refills += take;
  }
  *dry = f.finished && f.head == f.produced;
  if (want > 0 && !*dry) ++f.starved;
  return n_out;
}

inline void feed_release(Feed& f, const std::vector<FeedRange>& held) {
  if (held.empty()) return;
  {
    std::lock_guard<std::mutex> lk(f.mu);
    for (const FeedRange& r : held) f.chunk[r.seq % kStageChunks].released += r.n;
  }
  f.cv.notify_all();
}

inline void feed_fail(Feed& f) {
  {
    std::lock_guard<std::mutex> lk(f.mu);
    f.failed = true;
  }
  f.cv.notify_all();
}

inline bool feed_failed(Feed& f) {
  std::lock_guard<std::mutex> lk(f.mu);
  return f.failed;
}

// a group with nothing live sleeps here until the encoder delivers (or there is nothing left to wait for)
inline void feed_wait(Feed& f) {
  std::unique_lock<std::mutex> lk(f.mu);
  f.cv.wait(lk, [&] {
    if (f.failed || f.finished) return true;
    for (int q = f.head; q < f.produced; ++q)
      if (f.chunk[q % kStageChunks].taken < f.chunk[q % kStageChunks].n) return true;
    return false;
  });
}

// ---- producer side
// Sequence q's turn: waits until the ring chunk q % kStageChunks (which still holds sequence q - kStageChunks) has been
// given back completely, then claims the next <= cap segments.  false: nothing left to encode, or somebody failed.
// *pad: segments in front of `first` to encode along (see StageChunk::pad).
inline bool feed_claim(Feed& f, int q, int cap, int min_batch, int* first, int* n, int* pad) {
  std::unique_lock<std::mutex> lk(f.mu);
  if (f.next_seg >= f.n_total) return false;
  StageChunk& ch = f.chunk[q % kStageChunks];
  f.cv.wait(lk, [&] { return f.failed || q < kStageChunks || ch.released == ch.n; });
  if (f.failed) return false;
  *first = f.next_seg;
  *n = f.n_total - *first < cap ? f.n_total - *first : cap;
  f.next_seg += *n;
  *pad = *n < min_batch ? min_batch - *n : 0;
  if (*pad > *first) *pad = *first;
  return true;
}

// the pass of sequence q has finished on the device: its chunk goes on offer
inline void feed_publish(Feed& f, int q, int first, int n, int pad) {
  {
    std::lock_guard<std::mutex> lk(f.mu);
    StageChunk& ch = f.chunk[q % kStageChunks];
    ch = StageChunk();
    ch.first_seg = first;
    ch.n = n;
    ch.pad = pad;
    ch.batch = pad + n;
    ++f.produced;
  }
  f.cv.notify_all();
}

inline void feed_finish(Feed& f, bool failed) {
  {
    std::lock_guard<std::mutex> lk(f.mu);
    f.finished = true;
    if (failed) f.failed = true;
  }
  f.cv.notify_all();
}

}  // namespace mt3feed
#endif  // MT3_FEED_H_
