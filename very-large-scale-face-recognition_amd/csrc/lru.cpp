// Host-side LRU slot allocator + Dynamic-Class-Pool bookkeeping (C-ABI sections 1 and 2 of
// include/vlsfr.h).  Behavioural contract: reference lru.py:21-255 and the Python loops of
// ffc.py:162-177, 189-192, 214-235, 242-245, 256-259.
//
// Design (not a translation of the reference's dict + linked Python objects): every live key owns
// exactly one pool slot, so list nodes ARE slots — prev/next/key are flat arrays indexed by slot,
// the two sentinels sit at indices capacity and capacity+1, and the key→slot map is an
// open-addressing table of int32 slots (the key is read back from key_of[slot]).  An undo record
// is self-contained (type, neighbours, slot, evicted key), so rollback never needs the evicted
// node object to survive.
#include "vlsfr.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <unordered_map>
#include <vector>

#include "common_host.h"

namespace {

enum OpType : int32_t { OP_ADD = 0, OP_OVERFLOW = 1, OP_GET = 2 };

struct UndoRec {
  int32_t type;
  int32_t prev;
  int32_t next;
  int32_t slot;
  int64_t old_key;
};

inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

}  // namespace

struct vlsfr_lru {
  int64_t capacity = 0;
  int64_t cur_idx = 0;   // next never-used slot (reference: LRU.cur_idx)
  int64_t live = 0;      // len(cache)
  int32_t head = 0, tail = 0;
  std::vector<int32_t> prev, next;
  std::vector<int64_t> key_of;
  std::vector<int32_t> table;  // -1 empty, else slot
  uint64_t tmask = 0;
  std::vector<UndoRec> ops;

  // ---- key → slot table (linear probing, backward-shift deletion) ----
  int32_t find(int64_t key) const {
    uint64_t i = mix64((uint64_t)key) & tmask;
    for (;;) {
      int32_t s = table[i];
      if (s < 0) return -1;
      if (key_of[s] == key) return s;
      i = (i + 1) & tmask;
    }
  }
  void insert(int64_t key, int32_t slot) {
    key_of[slot] = key;
    uint64_t i = mix64((uint64_t)key) & tmask;
    while (table[i] >= 0) i = (i + 1) & tmask;
    table[i] = slot;
    ++live;
  }
  void erase(int64_t key) {
    uint64_t i = mix64((uint64_t)key) & tmask;
    for (;;) {
      int32_t s = table[i];
      if (s < 0) return;  // not present
      if (key_of[s] == key) break;
      i = (i + 1) & tmask;
    }
    // backward-shift
    uint64_t j = i;
    for (;;) {
      j = (j + 1) & tmask;
      int32_t s = table[j];
      if (s < 0) break;
      uint64_t home = mix64((uint64_t)key_of[s]) & tmask;
      // can s move to i?  yes iff home is cyclically outside (i, j]
      bool movable = (i <= j) ? (home <= i || home > j) : (home <= i && home > j);
      if (movable) {
        table[i] = s;
        i = j;
      }
    }
    table[i] = -1;
    --live;
  }

  // ---- list surgery ----
  void unlink(int32_t n) {
    int32_t p = prev[n], q = next[n];
    next[p] = q;
    prev[q] = p;
  }
  void push_front(int32_t n) {
    int32_t f = next[head];
    next[n] = f;
    prev[f] = n;
    prev[n] = head;
    next[head] = n;
  }

  // lru.py:44-89 (get) and lru.py:157-204 (try_get): identical list effect; `record` adds the
  // undo record the transactional variant pushes.
  int32_t touch(int64_t key, bool record) {
    int32_t s = find(key);
    if (s >= 0) {
      if (record) ops.push_back({OP_GET, prev[s], next[s], s, 0});
      unlink(s);
      push_front(s);
      return s;
    }
    if (cur_idx < capacity) {
      int32_t r = (int32_t)cur_idx++;
      insert(key, r);
      int32_t old_first = next[head];
      push_front(r);
      if (record) ops.push_back({OP_ADD, head, old_first, r, 0});
      return r;
    }
    // full: evict the tail node and reuse its slot
    int32_t victim = prev[tail];
    int64_t old_key = key_of[victim];
    if (record) ops.push_back({OP_OVERFLOW, prev[victim], next[victim], victim, old_key});
    unlink(victim);
    erase(old_key);
    insert(key, victim);
    push_front(victim);
    return victim;
  }

  // lru.py:210-248
  void undo_one() {
    UndoRec r = ops.back();
    ops.pop_back();
    if (r.type == OP_ADD) {
      next[r.prev] = r.next;
      prev[r.next] = r.prev;
      erase(key_of[r.slot]);
      --cur_idx;
    } else if (r.type == OP_OVERFLOW) {
      int32_t fresh = next[head];  // the reference pops whatever is first (LIFO assumption)
      unlink(fresh);
      erase(key_of[fresh]);
      // re-materialise the evicted node in the recorded position
      next[r.prev] = r.slot;
      prev[r.slot] = r.prev;
      next[r.slot] = r.next;
      prev[r.next] = r.slot;
      insert(r.old_key, r.slot);
    } else {
      int32_t cur = next[head];
      int32_t after = next[cur];
      next[head] = after;
      prev[after] = head;
      next[r.prev] = cur;
      prev[cur] = r.prev;
      next[cur] = r.next;
      prev[r.next] = cur;
    }
  }
};

extern "C" {

int vlsfr_lru_create(int64_t capacity, vlsfr_lru** out) {
  if (!out || capacity <= 0 || capacity > 0x7ffffff0LL)
    return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_create: capacity must be in [1, 2^31-16)");
  vlsfr_lru* h = new (std::nothrow) vlsfr_lru();
  if (!h) return vlsfr::fail(VLSFR_ENOMEM, "vlsfr_lru_create: out of memory");
  try {
    h->capacity = capacity;
    h->head = (int32_t)capacity;
    h->tail = (int32_t)capacity + 1;
    h->prev.assign((size_t)capacity + 2, -1);
    h->next.assign((size_t)capacity + 2, -1);
    h->key_of.assign((size_t)capacity, 0);
    uint64_t t = 16;
    while (t < (uint64_t)capacity * 2) t <<= 1;
    h->table.assign(t, -1);
    h->tmask = t - 1;
    h->next[h->head] = h->tail;
    h->prev[h->tail] = h->head;
  } catch (const std::bad_alloc&) {
    delete h;
    return vlsfr::fail(VLSFR_ENOMEM, "vlsfr_lru_create: out of memory");
  }
  *out = h;
  return VLSFR_OK;
}

void vlsfr_lru_destroy(vlsfr_lru* h) { delete h; }

int vlsfr_lru_get(vlsfr_lru* h, int64_t key, int32_t* slot) {
  if (!h || !slot) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_get: null argument");
  *slot = h->touch(key, false);
  return VLSFR_OK;
}

int vlsfr_lru_try_get(vlsfr_lru* h, int64_t key, int32_t* slot) {
  if (!h || !slot) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_try_get: null argument");
  *slot = h->touch(key, true);
  return VLSFR_OK;
}

int vlsfr_lru_view(const vlsfr_lru* h, int64_t key, int32_t* slot) {
  if (!h || !slot) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_view: null argument");
  *slot = h->find(key);
  return VLSFR_OK;
}

int vlsfr_lru_contains(const vlsfr_lru* h, int64_t key) { return h && h->find(key) >= 0 ? 1 : 0; }

int vlsfr_lru_rollback(vlsfr_lru* h, int64_t steps, int64_t* undone) {
  if (!h) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_rollback: null handle");
  int64_t n = 0;
  while (n < steps && !h->ops.empty()) {
    h->undo_one();
    ++n;
  }
  if (undone) *undone = n;
  return VLSFR_OK;
}

int vlsfr_lru_state(const vlsfr_lru* h, int64_t* keys, int32_t* slots, int64_t n_max, int64_t* n) {
  if (!h || !n) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_state: null argument");
  int64_t cnt = 0;
  for (int32_t cur = h->next[h->head]; cur != h->tail; cur = h->next[cur]) {
    if (cnt < n_max && keys && slots) {
      keys[cnt] = h->key_of[cur];
      slots[cnt] = cur;
    }
    ++cnt;
  }
  *n = cnt;
  if (cnt > n_max && keys) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_state: output arrays too small");
  return VLSFR_OK;
}

int vlsfr_lru_restore(vlsfr_lru* h, const int64_t* keys, const int32_t* slots, int64_t n) {
  if (!h || (n > 0 && (!keys || !slots))) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_restore: null argument");
  if (n > h->capacity) return vlsfr::fail(VLSFR_ESTATE, "vlsfr_lru_restore: more entries than capacity");
  if (h->cur_idx != 0) return vlsfr::fail(VLSFR_ESTATE, "vlsfr_lru_restore: LRU is not empty (cur_idx != 0)");
  // validate before touching the list: slots in range and distinct (a slot is a node here); keys are checked
  // for duplicates while they go into the (empty: cur_idx == 0) key table, which is wiped again on failure —
  // no second hash map, so a 10 M-entry restore costs one pass
  {
    std::vector<uint8_t> seen((size_t)(n > 0 ? n : 1), 0);
    for (int64_t i = 0; i < n; ++i) {
      // a state_dict of a live LRU always holds exactly the slots 0..n-1 (handed out in order,
      // never freed); anything else would let a later Add collide with a restored slot
      if (slots[i] < 0 || slots[i] >= n)
        return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_restore: slots must be a permutation of 0..n-1");
      if (seen[slots[i]]) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_restore: duplicate slot");
      seen[slots[i]] = 1;
    }
    if (h->live != 0) return vlsfr::fail(VLSFR_ESTATE, "vlsfr_lru_restore: LRU is not empty");
    for (int64_t i = 0; i < n; ++i) {
      if (h->find(keys[i]) >= 0) {
        std::fill(h->table.begin(), h->table.end(), -1);
        h->live = 0;
        return vlsfr::fail(VLSFR_ESTATE, "vlsfr_lru_restore: duplicate key");
      }
      h->insert(keys[i], slots[i]);
    }
  }
  int32_t cur = h->head;
  for (int64_t i = 0; i < n; ++i) {
    int32_t s = slots[i];
    h->next[cur] = s;
    h->prev[s] = cur;
    cur = s;
    ++h->cur_idx;
  }
  h->next[cur] = h->tail;
  h->prev[h->tail] = cur;
  return VLSFR_OK;
}

int vlsfr_lru_clear(vlsfr_lru* h) {
  if (!h) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_lru_clear: null handle");
  std::fill(h->table.begin(), h->table.end(), -1);
  h->live = 0;
  h->next[h->head] = h->tail;
  h->prev[h->tail] = h->head;
  // cur_idx and the op stack are left as they are (lru.py:132-141)
  return VLSFR_OK;
}

int64_t vlsfr_lru_capacity(const vlsfr_lru* h) { return h ? h->capacity : -1; }
int64_t vlsfr_lru_cur_idx(const vlsfr_lru* h) { return h ? h->cur_idx : -1; }
int64_t vlsfr_lru_size(const vlsfr_lru* h) { return h ? h->live : -1; }
int64_t vlsfr_lru_op_depth(const vlsfr_lru* h) { return h ? (int64_t)h->ops.size() : -1; }
int vlsfr_lru_op_type(const vlsfr_lru* h, int64_t i) {
  if (!h || i < 0 || i >= (int64_t)h->ops.size()) return VLSFR_EINVAL;
  return h->ops[(size_t)i].type;
}

// ------------------------------------------------------------------------------------------
// DCP bookkeeping for one FFC pass.
// ------------------------------------------------------------------------------------------
int vlsfr_dcp_assign(vlsfr_lru* h, uint8_t* qp, const int64_t* gallery_label, const int64_t* probe_label,
                     int32_t n, int transactional, int32_t* rows, int32_t* cols, int32_t* pool_label,
                     int32_t* ones_idx, int32_t* special_col, int32_t* src1, int32_t* src2,
                     int32_t* undo_slot, uint8_t* undo_val, vlsfr_dcp_plan* plan) {
  if (!h || !qp || !gallery_label || !probe_label || !rows || !cols || !pool_label || !ones_idx ||
      !special_col || !src1 || !src2 || !plan || n < 0)
    return vlsfr::fail(VLSFR_EINVAL, "vlsfr_dcp_assign: null argument");
  if (transactional && (!undo_slot || !undo_val))
    return vlsfr::fail(VLSFR_EINVAL, "vlsfr_dcp_assign: transactional pass needs undo buffers");

  std::unordered_map<int32_t, int32_t> writer[2];  // slot -> last gallery row writing (row r, slot)
  std::unordered_map<int32_t, int32_t> special;    // slot -> index in special_col
  std::unordered_map<int32_t, uint8_t> saved;      // transactional: slots whose qp was saved
  std::unordered_map<int32_t, uint8_t> in_ones;
  writer[0].reserve((size_t)n * 2);
  writer[1].reserve((size_t)n * 2);
  special.reserve((size_t)n * 6);
  int32_t n_ones = 0, n_special = 0, n_undo = 0;

  auto mark_special = [&](int32_t slot) {
    if (special.emplace(slot, n_special).second) special_col[n_special++] = slot;
  };

  for (int32_t i = 0; i < n; ++i) {
    const int64_t gl = gallery_label[i];
    const bool known = h->find(gl) >= 0;          // ffc.py:167 / :220  `gl not in self.lru`
    const int32_t idx = h->touch(gl, transactional != 0);
    if (transactional && saved.emplace(idx, 1).second) {  // ffc.py:224-225, 229-230
      undo_slot[n_undo] = idx;
      undo_val[n_undo] = qp[idx];
      ++n_undo;
    }
    int32_t r;
    if (!known) {            // ffc.py:168-171 / 221-226
      r = 0;
      qp[idx] = 1;
    } else {                 // ffc.py:173-177 / 228-234
      r = qp[idx];
      if (in_ones.emplace(idx, 1).second) ones_idx[n_ones++] = idx;
      qp[idx] = (uint8_t)((qp[idx] + 1) & 1);
    }
    rows[i] = r;
    cols[i] = idx;
    writer[r][idx] = i;      // duplicates: highest batch index wins
    mark_special(idx);
  }
  int32_t n_pos = 0;
  for (int32_t i = 0; i < n; ++i) {  // ffc.py:189-194 / 242-246
    int32_t s = h->find(probe_label[i]);
    pool_label[i] = s;
    if (s >= 0) {
      ++n_pos;
      mark_special(s);
    }
  }
  for (int32_t k = 0; k < n_ones; ++k) mark_special(ones_idx[k]);
  for (int32_t k = 0; k < n_special; ++k) {
    const int32_t c = special_col[k];
    auto w0 = writer[0].find(c);
    const int32_t s1 = (w0 != writer[0].end()) ? w0->second : -1;
    int32_t s2 = s1;
    if (in_ones.count(c)) {  // ffc.py:198-200 / 250-252: mask row → queue[1]
      auto w1 = writer[1].find(c);
      s2 = (w1 != writer[1].end()) ? w1->second : -2;
    }
    src1[k] = s1;
    src2[k] = s2;
  }
  plan->n = n;
  plan->n_ones = n_ones;
  plan->n_special = n_special;
  plan->n_pos = n_pos;
  plan->n_undo = n_undo;
  plan->steps = transactional ? n : 0;
  return VLSFR_OK;
}

int vlsfr_dcp_undo(vlsfr_lru* h, uint8_t* qp, const int32_t* undo_slot, const uint8_t* undo_val,
                   const vlsfr_dcp_plan* plan) {
  if (!h || !qp || !plan || (plan->n_undo > 0 && (!undo_slot || !undo_val)))
    return vlsfr::fail(VLSFR_EINVAL, "vlsfr_dcp_undo: null argument");
  for (int32_t k = 0; k < plan->n_undo; ++k) qp[undo_slot[k]] = undo_val[k];  // ffc.py:256-257
  int64_t steps = plan->steps;
  while (steps-- > 0 && !h->ops.empty()) h->undo_one();                         // ffc.py:259
  return VLSFR_OK;
}

}  // extern "C"
