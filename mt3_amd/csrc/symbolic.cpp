// Host-side symbolic stage of the MT3 path: event tokens -> notes.
//
// Product code (not the oracle).  C++ re-design of what the reference does in
// pure Python per token:
//   event_codec.Codec.decode_event_index            mt3/event_codec.py:103-112
//   run_length_encoding.decode_events               mt3/run_length_encoding.py:371-423
//   note_sequences.decode_note_event / _onset_event mt3/note_sequences.py:284-387
//   note_sequences.flush_note_decoding_state        mt3/note_sequences.py:396-408
//   note_sequences.assign_instruments               mt3/note_sequences.py:72-84
//   metrics_utils.decode_and_combine_predictions    mt3/metrics_utils.py:59-116
// Design: one pass over a flat token buffer with a prefix-offset codec table, an
// insertion-ordered active-note list (the reference relies on Python dict order
// when it ends un-tied notes and when it flushes), and "invalid event" signalled
// by a bool instead of exceptions.  All times are doubles evaluated with the
// reference's own expressions (start + steps / steps_per_second), so they are
// bit-identical; build with -ffp-contract=off.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "mt3_hip.h"
#include "common.h"

namespace {

constexpr double kDefaultNoteDuration = 0.01;  // note_sequences.py:29
constexpr double kMinNoteDuration = 0.01;      // note_sequences.py:32
constexpr int kDefaultVelocity = 100;          // note_sequences.py:28
constexpr int kMaxMidiVelocity = 127;          // note_seq.MAX_MIDI_VELOCITY

struct CodecTable {
  double steps_per_second = 100.0;
  int n = 0;
  int type[8], lo[8], hi[8], start[9];
  int velocity_bins = 0;
  bool ok = false;

  explicit CodecTable(const mt3_codec* c) {
    if (!c || c->num_ranges < 1 || c->num_ranges > 8) return;
    if (c->ranges[0].type != MT3_EV_SHIFT || c->ranges[0].min_value != 0) return;
    n = c->num_ranges;
    steps_per_second = c->steps_per_second;
    int off = 0;
    for (int i = 0; i < n; ++i) {
      type[i] = c->ranges[i].type;
      lo[i] = c->ranges[i].min_value;
      hi[i] = c->ranges[i].max_value;
      if (hi[i] < lo[i]) return;
      for (int j = 0; j < i; ++j)
        if (type[j] == type[i]) return;  // event types must be unique (event_codec.py:59-61)
      start[i] = off;
      off += hi[i] - lo[i] + 1;
      if (type[i] == MT3_EV_VELOCITY) velocity_bins = hi[i] - lo[i];  // vocabularies.py:57-60
    }
    start[n] = off;
    ok = true;
  }
  int num_classes() const { return start[n]; }
  // false == the reference's ValueError('Unknown event index')
  bool decode(int64_t index, int* t, int* v) const {
    if (index < 0 || index >= start[n]) return false;
    int k = 0;
    while (index >= start[k + 1]) ++k;
    *t = type[k];
    *v = lo[k] + static_cast<int>(index - start[k]);
    return true;
  }
  bool encode(int t, int v, int* index) const {
    for (int k = 0; k < n; ++k)
      if (type[k] == t) {
        if (v < lo[k] || v > hi[k]) return false;
        *index = start[k] + v - lo[k];
        return true;
      }
    return false;
  }
};

struct Active {
  int pitch, program;
  double onset;
  int velocity;
};

class NoteMachine {
 public:
  NoteMachine(int spec, const CodecTable& codec) : spec_(spec), codec_(codec) {}

  void begin_segment() {
    if (spec_ == MT3_SPEC_TIES) {  // begin_tied_pitches_section
      tied_.clear();
      in_tie_section_ = true;
    }
  }

  // returns false where the reference raises ValueError (caller counts it invalid)
  bool consume(double time, int type, int value) {
    if (spec_ == MT3_SPEC_ONSETS) {
      if (type != MT3_EV_PITCH) return false;
      mt3_note n{};
      n.start_time = time;
      n.end_time = time + kDefaultNoteDuration;
      n.pitch = value;
      n.velocity = kDefaultVelocity;
      notes.push_back(n);
      total_time = std::max(total_time, time + kDefaultNoteDuration);
      return true;
    }
    if (time < current_time_) return false;
    current_time_ = time;
    switch (type) {
      case MT3_EV_PITCH: {
        const int prog = current_program_;
        const int at = find(value, prog);
        if (in_tie_section_) {
          if (at < 0) return false;
          for (const auto& t : tied_)
            if (t.first == value && t.second == prog) return false;
          tied_.emplace_back(value, prog);
        } else if (current_velocity_ == 0) {
          if (at < 0) return false;
          const Active a = active_[at];
          active_.erase(active_.begin() + at);
          emit(a.onset, time, value, a.velocity, prog, false);
        } else {
          if (at >= 0) {  // re-onset of a sounding note: end it, start a new one
            const Active a = active_[at];
            active_.erase(active_.begin() + at);
            emit(a.onset, time, value, a.velocity, prog, false);
          }
          active_.push_back(Active{value, prog, time, current_velocity_});
        }
        return true;
      }
      case MT3_EV_DRUM:
        if (current_velocity_ == 0) return false;
        emit(time, time + kDefaultNoteDuration, value, current_velocity_, 0, true);
        return true;
      case MT3_EV_VELOCITY:
        // vocabularies.bin_to_velocity: int(127 * bin / num_bins), 0 stays 0
        current_velocity_ =
            value == 0 ? 0
                       : static_cast<int>(static_cast<double>(kMaxMidiVelocity * value) /
                                          static_cast<double>(codec_.velocity_bins));
        return true;
      case MT3_EV_PROGRAM:
        current_program_ = value;
        return true;
      case MT3_EV_TIE: {
        if (!in_tie_section_) return false;
        std::vector<Active> keep;
        for (const Active& a : active_) {
          bool is_tied = false;
          for (const auto& t : tied_) is_tied |= (t.first == a.pitch && t.second == a.program);
          if (is_tied) keep.push_back(a);
          else emit(a.onset, current_time_, a.pitch, a.velocity, a.program, false);
        }
        active_.swap(keep);
        in_tie_section_ = false;
        return true;
      }
      default:
        return false;
    }
  }

  void flush() {
    if (spec_ == MT3_SPEC_ONSETS) return;  // NoteOnsetEncodingSpec: flush = identity
    for (const Active& a : active_) current_time_ = std::max(current_time_, a.onset + kMinNoteDuration);
    for (const Active& a : active_) emit(a.onset, current_time_, a.pitch, a.velocity, a.program, false);
    active_.clear();
    // assign_instruments: order of first appearance of a program, skipping 9
    std::vector<std::pair<int, int>> seen;
    for (mt3_note& n : notes) {
      if (n.is_drum) { n.instrument = 9; continue; }
      int found = -1;
      for (const auto& s : seen)
        if (s.first == n.program) found = s.second;
      if (found < 0) {
        const int k = static_cast<int>(seen.size());
        found = k < 9 ? k : k + 1;
        seen.emplace_back(n.program, found);
      }
      n.instrument = found;
    }
  }

  std::vector<mt3_note> notes;
  double total_time = 0.0;

 private:
  int find(int pitch, int program) const {
    for (size_t i = 0; i < active_.size(); ++i)
      if (active_[i].pitch == pitch && active_[i].program == program) return static_cast<int>(i);
    return -1;
  }
  void emit(double start, double end, int pitch, int velocity, int program, bool drum) {
    end = std::max(end, start + kMinNoteDuration);
    mt3_note n{};
    n.start_time = start;
    n.end_time = end;
    n.pitch = pitch;
    n.velocity = velocity;
    n.program = program;
    n.is_drum = drum ? 1 : 0;
    notes.push_back(n);
    total_time = std::max(total_time, end);
  }

  int spec_;
  const CodecTable& codec_;
  double current_time_ = 0.0;
  int current_velocity_ = kDefaultVelocity;
  int current_program_ = 0;
  std::vector<Active> active_;              // insertion ordered, like the reference's dict
  std::vector<std::pair<int, int>> tied_;
  bool in_tie_section_ = false;
};

}  // namespace

extern "C" {

int mt3_build_codec(int32_t steps_per_second, int32_t max_shift_seconds, int32_t num_velocity_bins,
                    mt3_codec* out) {
  if (!out || steps_per_second <= 0 || max_shift_seconds <= 0 || num_velocity_bins < 1)
    return mt3::fail(MT3_ERR_INVALID, "mt3_build_codec: bad arguments");
  std::memset(out, 0, sizeof(*out));
  out->steps_per_second = steps_per_second;
  out->num_ranges = 6;
  const mt3_event_range r[6] = {
      {MT3_EV_SHIFT, 0, steps_per_second * max_shift_seconds},
      {MT3_EV_PITCH, 0, 127},
      {MT3_EV_VELOCITY, 0, num_velocity_bins},
      {MT3_EV_TIE, 0, 0},
      {MT3_EV_PROGRAM, 0, 127},
      {MT3_EV_DRUM, 0, 127}};
  for (int i = 0; i < 6; ++i) out->ranges[i] = r[i];
  return MT3_OK;
}

int mt3_codec_num_classes(const mt3_codec* c) {
  CodecTable t(c);
  if (!t.ok) return mt3::fail(MT3_ERR_INVALID, "mt3_codec_num_classes: malformed codec");
  return t.num_classes();
}

int mt3_codec_decode_event(const mt3_codec* c, int32_t index, int32_t* type, int32_t* value) {
  CodecTable t(c);
  if (!t.ok || !type || !value) return mt3::fail(MT3_ERR_INVALID, "mt3_codec_decode_event: malformed codec");
  int ty, v;
  if (!t.decode(index, &ty, &v)) return mt3::fail(MT3_ERR_INVALID, "Unknown event index");
  *type = ty;
  *value = v;
  return MT3_OK;
}

int mt3_codec_encode_event(const mt3_codec* c, int32_t type, int32_t value, int32_t* index) {
  CodecTable t(c);
  if (!t.ok || !index) return mt3::fail(MT3_ERR_INVALID, "mt3_codec_encode_event: malformed codec");
  int idx;
  if (!t.encode(type, value, &idx)) return mt3::fail(MT3_ERR_INVALID, "event type/value not in codec");
  *index = idx;
  return MT3_OK;
}

int mt3_notes_decode(const mt3_codec* c, int32_t spec, int32_t n_segments, const int32_t* h_tokens,
                     const int64_t* h_seg_offsets, const double* h_start_times,
                     const int32_t* h_has_max_time, const double* h_max_times,
                     mt3_note* h_notes, int64_t notes_capacity, int64_t* n_notes,
                     int64_t* invalid_events, int64_t* dropped_events, double* total_time) {
  CodecTable codec(c);
  if (!codec.ok) return mt3::fail(MT3_ERR_INVALID, "mt3_notes_decode: malformed codec");
  if (spec < MT3_SPEC_ONSETS || spec > MT3_SPEC_TIES || n_segments < 0 || !n_notes)
    return mt3::fail(MT3_ERR_INVALID, "mt3_notes_decode: bad spec / arguments");
  if (n_segments > 0 && (!h_seg_offsets || !h_start_times))
    return mt3::fail(MT3_ERR_INVALID, "mt3_notes_decode: null segment arrays");

  // sorted(predictions, key=start_time): stable
  std::vector<int> order(n_segments);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](int a, int b) { return h_start_times[a] < h_start_times[b]; });

  NoteMachine m(spec, codec);
  int64_t invalid = 0, dropped = 0;
  for (int k = 0; k < n_segments; ++k) {
    const int s = order[k];
    m.begin_segment();
    bool has_max;
    double max_time = 0.0;
    if (h_has_max_time) {
      has_max = h_has_max_time[s] != 0;
      if (has_max) max_time = h_max_times[s];
    } else {
      has_max = k + 1 < n_segments;
      if (has_max) max_time = h_start_times[order[k + 1]];
    }
    const double start = h_start_times[s];
    const int64_t b = h_seg_offsets[s], e = h_seg_offsets[s + 1];
    if (e < b) return mt3::fail(MT3_ERR_INVALID, "mt3_notes_decode: seg_offsets not monotone");
    int64_t steps = 0;
    double now = start;
    for (int64_t i = b; i < e; ++i) {
      int type, value;
      if (!codec.decode(h_tokens[i], &type, &value)) { ++invalid; continue; }
      if (type == MT3_EV_SHIFT) {
        steps += value;
        now = start + static_cast<double>(steps) / codec.steps_per_second;
        // `if max_time and cur_time > max_time`: None and 0.0 are both falsy
        if (has_max && max_time != 0.0 && now > max_time) { dropped += e - i; break; }
      } else {
        steps = 0;
        if (!m.consume(now, type, value)) ++invalid;
      }
    }
  }
  m.flush();
  *n_notes = static_cast<int64_t>(m.notes.size());
  if (invalid_events) *invalid_events = invalid;
  if (dropped_events) *dropped_events = dropped;
  if (total_time) *total_time = m.total_time;
  if (*n_notes > notes_capacity) return mt3::fail(MT3_ERR_CAPACITY, "mt3_notes_decode: notes buffer too small");
  if (*n_notes > 0) {
    if (!h_notes) return mt3::fail(MT3_ERR_INVALID, "mt3_notes_decode: null notes buffer");
    std::memcpy(h_notes, m.notes.data(), sizeof(mt3_note) * m.notes.size());
  }
  return MT3_OK;
}

}  // extern "C"
