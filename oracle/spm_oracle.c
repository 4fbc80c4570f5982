/* TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-thread CPU restatement of the reference's encode hot path:
 *     SentencePieceProcessor::Encode(input, vector<int>*)
 *       = Normalizer::Normalize -> {unigram EncodeOptimized | BPE SampleEncode(alpha 0)}
 *         -> PopulateSentencePieceText (ids only) -> ApplyExtraOptions.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * it, and only as the checker.  The product (sentencepiece_amd/) never links it.
 *
 * Parity is PINNED: tests/test_oracle.py checks this file against (a) the
 * golden ids in tests/golden/ produced by the reference compiled from
 * /root/reference (oracle/_ref), incl. the survey's botchan KAT (95,515 ids,
 * md5 ff197d02...), and (b) oracle/_ref itself where that library is present.
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference).  Data structures are deliberately naive (a linked trie, a
 * linear-probe string hash, a binary heap); nothing here is tuned.
 */
#define _POSIX_C_SOURCE 200809L
#include "spm_oracle.h"

#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ utils */
typedef struct { const uint8_t *p; size_t n; } sv;

static int sv_eq(sv a, sv b) { return a.n == b.n && (a.n == 0 || memcmp(a.p, b.p, a.n) == 0); }
static int sv_eq_c(sv a, const char *c) { size_t n = strlen(c); return a.n == n && memcmp(a.p, c, n) == 0; }

enum { T_NORMAL = 1, T_UNKNOWN = 2, T_CONTROL = 3, T_USER_DEFINED = 4, T_UNUSED = 5, T_BYTE = 6 };
enum { M_UNIGRAM = 1, M_BPE = 2, M_WORD = 3, M_CHAR = 4 };
enum { OPT_REVERSE = 0, OPT_BOS = 1, OPT_EOS = 2 };

typedef struct { sv piece; float score; int type; } Piece;

/* linked trie: children of a node are a sibling list; root gets a table. */
typedef struct {
  int *first_child, *sibling, *value;
  uint8_t *label;
  int n, cap;
  int root_child[256];
} Trie;

typedef struct { sv key; int val; } HEnt;
typedef struct { HEnt *e; size_t cap; } SMap; /* string -> int, val -1 == empty */

struct Oracle {
  uint8_t *buf; size_t buf_n;           /* private copy of the model bytes */
  Piece *pieces; int n_pieces;
  /* trainer_spec */
  int model_type, byte_fallback, ws_suffix;
  sv unk_piece, bos_piece, eos_piece, pad_piece;
  sv unk_surface; int has_unk_surface;   /* trainer_spec.unk_surface (sentencepiece_model.proto:228) */
  int has_denormalizer;                   /* denormalizer_spec with a charsmap (sentencepiece_processor.cc:248-252) */
  /* normalizer_spec */
  sv charsmap; int add_dummy_prefix, remove_extra_ws, escape_ws;
  const uint32_t *nunits; size_t n_nunits; const char *nstrings; size_t nstrings_n;
  /* model_interface state (model_interface.cc:63-151) */
  SMap pieces_map, reserved_map;
  int unk_id;
  Trie uds;   int has_uds;      /* PrefixMatcher over USER_DEFINED pieces */
  /* unigram (unigram_model.cc:652-670) */
  Trie ptrie; float min_score, max_score;
  int byte_ids[256];
  int opts[16]; int n_opts;
  int opt_unk_piece;                 /* UNK_PIECE ("unk" / "unk_piece"): piece strings only */
};

/* ------------------------------------------------------------- tiny proto */
typedef struct { const uint8_t *p, *end; int err; } PB;

static uint64_t pb_varint(PB *b) {
  uint64_t v = 0; int sh = 0;
  while (b->p < b->end) {
    uint8_t c = *b->p++;
    v |= (uint64_t)(c & 0x7F) << sh;
    if (!(c & 0x80)) return v;
    sh += 7;
    if (sh > 63) break;
  }
  b->err = 1; return 0;
}
/* returns field number, fills wire type / payload. 0 at end. */
static int pb_next(PB *b, int *wt, uint64_t *val, sv *bytes) {
  if (b->err || b->p >= b->end) return 0;
  uint64_t key = pb_varint(b);
  if (b->err) return 0;
  *wt = (int)(key & 7);
  switch (*wt) {
    case 0: *val = pb_varint(b); break;
    case 1: if (b->end - b->p < 8) { b->err = 1; return 0; } memcpy(val, b->p, 8); b->p += 8; break;
    case 5: { if (b->end - b->p < 4) { b->err = 1; return 0; } uint32_t t; memcpy(&t, b->p, 4); *val = t; b->p += 4; break; }
    case 2: {
      uint64_t n = pb_varint(b);
      if (b->err || (uint64_t)(b->end - b->p) < n) { b->err = 1; return 0; }
      bytes->p = b->p; bytes->n = (size_t)n; b->p += n; break;
    }
    default: b->err = 1; return 0;
  }
  return (int)(key >> 3);
}

/* ------------------------------------------------------------------ trie */
static void trie_init(Trie *t) {
  memset(t, 0, sizeof(*t));
  t->cap = 1024; t->n = 1;
  t->first_child = malloc(sizeof(int) * t->cap); t->sibling = malloc(sizeof(int) * t->cap);
  t->value = malloc(sizeof(int) * t->cap); t->label = malloc(t->cap);
  t->first_child[0] = -1; t->sibling[0] = -1; t->value[0] = -1; t->label[0] = 0;
  for (int i = 0; i < 256; ++i) t->root_child[i] = -1;
}
static void trie_free(Trie *t) { free(t->first_child); free(t->sibling); free(t->value); free(t->label); }
static int trie_child(const Trie *t, int node, uint8_t c) {
  if (node == 0) return t->root_child[c];
  for (int k = t->first_child[node]; k >= 0; k = t->sibling[k]) if (t->label[k] == c) return k;
  return -1;
}
static void trie_insert(Trie *t, sv key, int value) {
  int node = 0;
  for (size_t i = 0; i < key.n; ++i) {
    int k = trie_child(t, node, key.p[i]);
    if (k < 0) {
      if (t->n == t->cap) {
        t->cap *= 2;
        t->first_child = realloc(t->first_child, sizeof(int) * t->cap);
        t->sibling = realloc(t->sibling, sizeof(int) * t->cap);
        t->value = realloc(t->value, sizeof(int) * t->cap);
        t->label = realloc(t->label, t->cap);
      }
      k = t->n++;
      t->first_child[k] = -1; t->value[k] = -1; t->label[k] = key.p[i];
      t->sibling[k] = t->first_child[node]; t->first_child[node] = k;
      if (node == 0) t->root_child[key.p[i]] = k;
    }
    node = k;
  }
  t->value[node] = value;
}

/* --------------------------------------------------------- string -> int */
static uint64_t fnv(sv s) { uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < s.n; ++i) { h ^= s.p[i]; h *= 1099511628211ull; } return h; }
static void smap_init(SMap *m, size_t n) {
  m->cap = 16; while (m->cap < 2 * n + 8) m->cap *= 2;
  m->e = malloc(sizeof(HEnt) * m->cap);
  for (size_t i = 0; i < m->cap; ++i) m->e[i].val = -1;
}
static int smap_find(const SMap *m, sv k) {
  if (!m->e) return -1;
  for (size_t i = fnv(k) & (m->cap - 1);; i = (i + 1) & (m->cap - 1)) {
    if (m->e[i].val < 0) return -1;
    if (sv_eq(m->e[i].key, k)) return m->e[i].val;
  }
}
static int smap_insert(SMap *m, sv k, int v) { /* 0 if already present */
  for (size_t i = fnv(k) & (m->cap - 1);; i = (i + 1) & (m->cap - 1)) {
    if (m->e[i].val < 0) { m->e[i].key = k; m->e[i].val = v; return 1; }
    if (sv_eq(m->e[i].key, k)) return 0;
  }
}

/* -------------------------------------------------- utf-8 (util.h / .cc) */
/* util.h:151-153 */
static int one_char_len(uint8_t c) { return "\1\1\1\1\1\1\1\1\1\1\1\1\2\2\3\4"[c >> 4]; }
static int is_trail(uint8_t x) { return (int8_t)x < -0x40; }                         /* util.h:157 */
static int valid_cp(uint32_t c) { return c < 0xD800 || (c >= 0xE000 && c <= 0x10FFFF); } /* util.h:159-161 */
/* util.cc:51-84; returns code point, 0xFFFD + mblen 1 on error */
static uint32_t decode_utf8(const uint8_t *b, size_t len, size_t *mblen) {
  if (b[0] < 0x80) { *mblen = 1; return b[0]; }
  if (len >= 2 && (b[0] & 0xE0) == 0xC0) {
    uint32_t cp = ((b[0] & 0x1Fu) << 6) | (b[1] & 0x3Fu);
    if (is_trail(b[1]) && cp >= 0x80 && valid_cp(cp)) { *mblen = 2; return cp; }
  } else if (len >= 3 && (b[0] & 0xF0) == 0xE0) {
    uint32_t cp = ((b[0] & 0x0Fu) << 12) | ((b[1] & 0x3Fu) << 6) | (b[2] & 0x3Fu);
    if (is_trail(b[1]) && is_trail(b[2]) && cp >= 0x800 && valid_cp(cp)) { *mblen = 3; return cp; }
  } else if (len >= 4 && (b[0] & 0xF8) == 0xF0) {
    uint32_t cp = ((b[0] & 0x07u) << 18) | ((b[1] & 0x3Fu) << 12) | ((b[2] & 0x3Fu) << 6) | (b[3] & 0x3Fu);
    if (is_trail(b[1]) && is_trail(b[2]) && is_trail(b[3]) && cp >= 0x10000 && valid_cp(cp)) { *mblen = 4; return cp; }
  }
  *mblen = 1; return 0xFFFD;
}
/* util.h:173-176 */
static int is_valid_decode_utf8(const uint8_t *b, size_t len, size_t *mblen) {
  uint32_t c = decode_utf8(b, len, mblen);
  return c != 0xFFFD || *mblen == 3;
}

/* --------------------------------------------- Darts units (darts.h:50-80) */
static uint32_t du_offset(uint32_t u) { return (u >> 10) << ((u & (1u << 9)) >> 6); }
static uint32_t du_label(uint32_t u) { return u & ((1u << 31) | 0xFF); }
static int du_has_leaf(uint32_t u) { return (u >> 8) & 1; }
static uint32_t du_value(uint32_t u) { return u & ((1u << 31) - 1); }

/* PrefixMatcher::PrefixMatch (normalizer.cc:324-346) over USER_DEFINED pieces */
static int prefix_match(const Oracle *o, const uint8_t *w, size_t n, int *found) {
  int mblen = 0;
  if (o->has_uds) {
    int node = 0;
    for (size_t i = 0; i < n; ++i) {
      node = trie_child(&o->uds, node, w[i]);
      if (node < 0) break;
      if (o->uds.value[node] >= 0) mblen = (int)(i + 1);
    }
  }
  if (found) *found = mblen > 0;
  if (mblen > 0) return mblen;
  int l = one_char_len(w[0]);
  return (int)n < l ? (int)n : l;
}

/* Normalizer::NormalizePrefix (normalizer.cc:195-253) */
static size_t normalize_prefix(const Oracle *o, const uint8_t *in, size_t n, sv *out) {
  out->p = NULL; out->n = 0;
  if (n == 0) return 0;
  if (o->has_uds) {
    int found = 0;
    int mblen = prefix_match(o, in, n, &found);
    if (found) { out->p = in; out->n = (size_t)mblen; return (size_t)mblen; }
  }
  size_t longest_length = 0; uint32_t longest_value = 0;
  if (o->nunits) {
    /* commonPrefixSearch (darts.h:467-513) + longest-rule select (normalizer.cc:222-228) */
    size_t pos = 0;
    uint32_t unit = o->nunits[pos];
    pos ^= du_offset(unit);
    for (size_t i = 0; i < n; ++i) {
      pos ^= in[i];
      if (pos >= o->n_nunits) break;
      unit = o->nunits[pos];
      if (du_label(unit) != in[i]) break;
      pos ^= du_offset(unit);
      if (du_has_leaf(unit)) {
        if (pos >= o->n_nunits) break;
        if (longest_length == 0 || i + 1 > longest_length) { longest_length = i + 1; longest_value = du_value(o->nunits[pos]); }
      }
    }
  }
  if (longest_length == 0) {
    size_t length = 0;
    if (!is_valid_decode_utf8(in, n, &length)) {
      static const uint8_t kRepl[] = {0xEF, 0xBF, 0xBD};
      out->p = kRepl; out->n = 3; return 1;
    }
    out->p = in; out->n = length; return length;
  }
  out->p = (const uint8_t *)o->nstrings + longest_value;
  out->n = strnlen(o->nstrings + longest_value, o->nstrings_n - longest_value);
  return longest_length;
}

typedef struct { uint8_t *p; size_t n, cap; } Buf;
static void buf_push(Buf *b, const void *src, size_t n) {
  if (b->n + n > b->cap) { while (b->n + n > b->cap) b->cap = b->cap ? b->cap * 2 : 256; b->p = realloc(b->p, b->cap); }
  memcpy(b->p + b->n, src, n); b->n += n;
}

/* norm_to_orig (normalizer.cc:73): one uint32 per normalized byte + the closing entry */
typedef struct { uint32_t *p; size_t n, cap; } Align;
static void align_push(Align *a, uint32_t v, size_t times) {
  if (!a) return;
  for (size_t i = 0; i < times; ++i) {
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 256; a->p = realloc(a->p, sizeof(uint32_t) * a->cap); }
    a->p[a->n++] = v;
  }
}

/* Normalizer::Normalize (normalizer.cc:71-186); n2o (optional) receives the alignment vector. */
static void normalize_aligned(const Oracle *o, const uint8_t *in, size_t n, Buf *out, Align *n2o) {
  static const uint8_t kSpaceSymbol[] = {0xE2, 0x96, 0x81};
  out->n = 0;
  if (n2o) n2o->n = 0;
  if (n == 0) return;
  sv sp;
  uint32_t consumed = 0;                                                                /* :83 */
  if (o->remove_extra_ws) {
    while (n > 0) {
      size_t c = normalize_prefix(o, in, n, &sp);
      if (!(sp.n == 1 && sp.p[0] == ' ')) break;
      in += c; n -= c; consumed += (uint32_t)c;                                         /* :92-93 */
    }
  }
  if (n == 0) return;
  if (!o->ws_suffix && o->add_dummy_prefix) {
    if (o->escape_ws) { buf_push(out, kSpaceSymbol, 3); align_push(n2o, consumed, 3); }  /* :109-120 */
    else { buf_push(out, " ", 1); align_push(n2o, consumed, 1); }
  }
  int is_prev_space = o->remove_extra_ws;
  while (n > 0) {
    size_t c = normalize_prefix(o, in, n, &sp);
    while (is_prev_space && sp.n > 0 && sp.p[0] == ' ') { sp.p++; sp.n--; }
    if (sp.n > 0) {
      for (size_t k = 0; k < sp.n; ++k) {
        if (o->escape_ws && sp.p[k] == ' ') { buf_push(out, kSpaceSymbol, 3); align_push(n2o, consumed, 3); }   /* :143-148 */
        else { buf_push(out, sp.p + k, 1); align_push(n2o, consumed, 1); }                                         /* :150-151 */
      }
      is_prev_space = sp.p[sp.n - 1] == ' ';
    }
    in += c; n -= c; consumed += (uint32_t)c;                                           /* :158 */
    if (!o->remove_extra_ws) is_prev_space = 0;
  }
  if (o->remove_extra_ws) {
    const uint8_t *space = o->escape_ws ? kSpaceSymbol : (const uint8_t *)" ";
    size_t sl = o->escape_ws ? 3 : 1;
    while (out->n >= sl && memcmp(out->p + out->n - sl, space, sl) == 0) {
      out->n -= sl;
      if (n2o) { consumed = n2o->p[out->n]; n2o->n = out->n; }                          /* :172-174 */
    }
  }
  if (o->ws_suffix && o->add_dummy_prefix) {
    if (o->escape_ws) { buf_push(out, kSpaceSymbol, 3); align_push(n2o, consumed, 3); }
    else { buf_push(out, " ", 1); align_push(n2o, consumed, 1); }
  }
  align_push(n2o, consumed, 1);                                                         /* :181 */
}
static void normalize(const Oracle *o, const uint8_t *in, size_t n, Buf *out) { normalize_aligned(o, in, n, out, NULL); }

/* --------------------------------------------------------------- results */
typedef struct { int begin, len, id; } Tok;   /* piece = normalized[begin, begin+len) */
typedef struct { Tok *p; size_t n, cap; } Toks;
static void toks_push(Toks *t, int begin, int len, int id) {
  if (t->n == t->cap) { t->cap = t->cap ? t->cap * 2 : 64; t->p = realloc(t->p, sizeof(Tok) * t->cap); }
  t->p[t->n].begin = begin; t->p[t->n].len = len; t->p[t->n].id = id; t->n++;
}

/* unigram::Model::EncodeOptimized (unigram_model.cc:889-1020).  The score
 * arithmetic mirrors the C++ types exactly: a piece candidate is a DOUBLE sum
 * compared against the FLOAT-stored best and stored rounded to float
 * (:979-989); the UNK candidate is pure float (:997-1001). */
static void unigram_encode(const Oracle *o, const uint8_t *norm, int size, Toks *out) {
  typedef struct { int id; float best_path_score; int starts_at; } Node;
  out->n = 0;
  if (size == 0) return;
  const float unk_score = o->min_score - 10.0f;           /* :955, kUnkPenalty :39 */
  Node *best = malloc(sizeof(Node) * (size_t)(size + 1));
  for (int i = 0; i <= size; ++i) { best[i].id = -1; best[i].best_path_score = 0; best[i].starts_at = -1; }
  int starts_at = 0;
  while (starts_at < size) {
    int node = 0;                                          /* trie_->traverse one byte at a time :970-971 */
    int key_pos = starts_at;
    const float best_till_here = best[starts_at].best_path_score;
    int has_single_node = 0;
    int mblen = one_char_len(norm[starts_at]);
    if (mblen > size - starts_at) mblen = size - starts_at;
    while (key_pos < size) {
      node = trie_child(&o->ptrie, node, norm[key_pos]);
      if (node < 0) break;                                 /* ret == -2 */
      ++key_pos;
      const int ret = o->ptrie.value[node];
      if (ret >= 0) {
        if (o->pieces[ret].type == T_UNUSED) continue;     /* :974 */
        Node *t = &best[key_pos];
        const size_t length = (size_t)(key_pos - starts_at);
        double score;
        if (o->pieces[ret].type == T_USER_DEFINED) {       /* :979-981 */
          const float prod = (float)length * o->max_score;
          score = (double)prod - 0.1;
        } else {
          score = (double)o->pieces[ret].score;
        }
        const double cand = score + (double)best_till_here;
        if (t->starts_at == -1 || cand > (double)t->best_path_score) {
          t->best_path_score = (float)cand; t->starts_at = starts_at; t->id = ret;
        }
        if (!has_single_node && length == (size_t)mblen) has_single_node = 1;
      }
    }
    if (!has_single_node) {                                /* :995-1005 */
      Node *t = &best[starts_at + mblen];
      const float cand = unk_score + best_till_here;
      if (t->starts_at == -1 || cand > t->best_path_score) {
        t->best_path_score = cand; t->starts_at = starts_at; t->id = o->unk_id;
      }
    }
    starts_at += mblen;
  }
  int ends_at = size;                                      /* :1010-1018 */
  while (ends_at > 0) {
    const Node *nd = &best[ends_at];
    toks_push(out, nd->starts_at, ends_at - nd->starts_at, nd->id);
    ends_at = nd->starts_at;
  }
  for (size_t i = 0, j = out->n ? out->n - 1 : 0; i < j; ++i, --j) { Tok t = out->p[i]; out->p[i] = out->p[j]; out->p[j] = t; }
  free(best);
}

/* ModelInterface::PieceToId (model_interface.cc:51-61) */
static int piece_to_id(const Oracle *o, sv piece) {
  int id = smap_find(&o->reserved_map, piece);
  if (id >= 0) return id;
  id = smap_find(&o->pieces_map, piece);
  if (id >= 0) return id;
  return o->unk_id;
}

/* bpe::Model::SampleEncode(normalized, alpha = 0) (bpe_model.cc:38-203). */
typedef struct { int left, right; float score; size_t size; } SymbolPair;
typedef struct { int prev, next, freeze; sv piece; } Symbol;
typedef struct { SymbolPair *p; size_t n, cap; } Heap;
/* SymbolPairComparator (bpe_model.cc:51-57): true if a sorts below b */
static int pair_less(const SymbolPair *a, const SymbolPair *b) {
  return a->score < b->score || (a->score == b->score && a->left > b->left);
}
static void heap_push(Heap *h, SymbolPair v) {
  if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 256; h->p = realloc(h->p, sizeof(SymbolPair) * h->cap); }
  size_t i = h->n++;
  while (i > 0) { size_t par = (i - 1) / 2; if (!pair_less(&h->p[par], &v)) break; h->p[i] = h->p[par]; i = par; }
  h->p[i] = v;
}
static SymbolPair heap_pop(Heap *h) {
  SymbolPair top = h->p[0], v = h->p[--h->n];
  size_t i = 0;
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m;
    if (l >= h->n) break;
    m = (r < h->n && pair_less(&h->p[l], &h->p[r])) ? r : l;
    if (!pair_less(&v, &h->p[m])) break;
    h->p[i] = h->p[m]; i = m;
  }
  if (h->n) h->p[i] = v;
  return top;
}

typedef struct { int set; size_t left_len; } RevMerge; /* keyed by piece id */

static void maybe_add_pair(const Oracle *o, Symbol *sym, Heap *agenda, RevMerge *rev, int left, int right) {
  if (left == -1 || right == -1 || sym[left].freeze || sym[right].freeze) return;   /* :85-87 */
  sv piece = { sym[left].piece.p, sym[left].piece.n + sym[right].piece.n };
  int id = smap_find(&o->pieces_map, piece);                                       /* :91 */
  if (id < 0) return;
  SymbolPair h = { left, right, o->pieces[id].score, piece.n };
  heap_push(agenda, h);
  if (o->pieces[id].type == T_UNUSED) { rev[id].set = 1; rev[id].left_len = sym[left].piece.n; }  /* :103-106 */
}

static void bpe_resegment(const Oracle *o, const RevMerge *rev, const uint8_t *norm, sv w, Toks *out) {
  const int id = piece_to_id(o, w);                                                 /* :178 */
  if (id == -1 || o->pieces[id].type != T_UNUSED) { toks_push(out, (int)(w.p - norm), (int)w.n, id); return; }
  if (!rev[id].set) { toks_push(out, (int)(w.p - norm), (int)w.n, id); return; }
  /* rev_merge maps the merged STRING to the (left, right) strings last
   * registered for it; those are a prefix/suffix split of w itself. */
  sv l = { w.p, rev[id].left_len }, r = { w.p + rev[id].left_len, w.n - rev[id].left_len };
  bpe_resegment(o, rev, norm, l, out);
  bpe_resegment(o, rev, norm, r, out);
}

static void bpe_encode(const Oracle *o, const uint8_t *norm, int size, Toks *out) {
  out->n = 0;
  if (size == 0) return;
  Symbol *sym = malloc(sizeof(Symbol) * (size_t)size);
  RevMerge *rev = calloc((size_t)o->n_pieces, sizeof(RevMerge));
  Heap agenda = {0};
  int n_sym = 0;
  const uint8_t *p = norm; size_t rem = (size_t)size;
  while (rem > 0) {                                                                 /* :109-120 */
    Symbol s; int found = 0;
    const int mblen = prefix_match(o, p, rem, &found);
    s.freeze = found; s.piece.p = p; s.piece.n = (size_t)mblen;
    s.prev = n_sym == 0 ? -1 : n_sym - 1;
    p += mblen; rem -= (size_t)mblen;
    s.next = rem == 0 ? -1 : n_sym + 1;
    sym[n_sym++] = s;
  }
  for (int i = 1; i < n_sym; ++i) maybe_add_pair(o, sym, &agenda, rev, i - 1, i);   /* :127-129 */
  while (agenda.n) {                                                                /* :142-173 */
    SymbolPair top = heap_pop(&agenda);
    if (sym[top.left].piece.n == 0 || sym[top.right].piece.n == 0 ||
        sym[top.left].piece.n + sym[top.right].piece.n != top.size) continue;
    sym[top.left].piece.n += sym[top.right].piece.n;
    sym[top.left].next = sym[top.right].next;
    if (sym[top.right].next >= 0) sym[sym[top.right].next].prev = top.left;
    sym[top.right].piece.n = 0;
    maybe_add_pair(o, sym, &agenda, rev, sym[top.left].prev, top.left);
    maybe_add_pair(o, sym, &agenda, rev, top.left, sym[top.left].next);
  }
  for (int i = 0; i != -1; i = sym[i].next) bpe_resegment(o, rev, norm, sym[i].piece, out);  /* :195-200 */
  free(sym); free(rev); free(agenda.p);
}

/* PopulateSentencePieceText, ids only (sentencepiece_processor.cc:547-636).
 * Returns -1 where the reference returns a non-OK status. */
typedef struct { int32_t *p; size_t n, cap; } Ids;
static void ids_push(Ids *v, int32_t x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->p = realloc(v->p, sizeof(int32_t) * v->cap); }
  v->p[v->n++] = x;
}
static void ids_insert_front(Ids *v, int32_t x) {
  ids_push(v, 0);
  memmove(v->p + 1, v->p, sizeof(int32_t) * (v->n - 1));
  v->p[0] = x;
}

static int populate_ids(const Oracle *o, const uint8_t *norm, int norm_size, const Toks *res, Ids *out) {
  size_t consumed = 0; int is_prev_unk = 0;
  out->n = 0;
  for (size_t k = 0; k < res->n; ++k) {
    const Tok *t = &res->p[k];
    if (t->len == 0) return -1;                             /* :557 */
    const int id = t->id;
    const int type = (id >= 0 && id < o->n_pieces) ? o->pieces[id].type : 0;
    const int is_unk = type == T_UNKNOWN;                   /* IsUnknown(id) */
    if (type == T_CONTROL) {                                /* :561-567, consumes nothing */
      ids_push(out, id);
    } else {
      if (is_unk && o->byte_fallback) {                     /* :581-603 one BYTE piece per byte */
        for (int i = 0; i < t->len; ++i) ids_push(out, o->byte_ids[norm[t->begin + i]]);
      } else if (is_prev_unk && is_unk) {                   /* :609-613 merged into the previous piece */
      } else {
        ids_push(out, id);
      }
      consumed += (size_t)t->len;
    }
    is_prev_unk = is_unk;
  }
  if (consumed != (size_t)norm_size) return -1;             /* :628 */
  /* ApplyExtraOptions (:1019-1064), in the order given */
  for (int k = 0; k < o->n_opts; ++k) {
    switch (o->opts[k]) {
      case OPT_REVERSE:
        for (size_t i = 0, j = out->n ? out->n - 1 : 0; i < j; ++i, --j) { int32_t t = out->p[i]; out->p[i] = out->p[j]; out->p[j] = t; }
        break;
      case OPT_EOS: { sv e = o->eos_piece; e.n = strnlen((const char *)e.p, e.n); ids_push(out, piece_to_id(o, e)); break; }
      case OPT_BOS: { sv b = o->bos_piece; b.n = strnlen((const char *)b.p, b.n); ids_insert_front(out, piece_to_id(o, b)); break; }
    }
  }
  return 0;
}

/* PopulateSentencePieceText with the spans (sentencepiece_processor.cc:547-636): pieces(i).id / begin / end.
 * input_size = bytes of the raw sentence (the bos / eos spans, :1029-1048). */
typedef struct { int32_t id; uint32_t begin, end; uint32_t nb, ne; int lit; } Span;   /* piece = normalized[nb, ne), or the
                                                                                        piece name of id when lit */
typedef struct { Span *p; size_t n, cap; } Spans;
static void spans_push(Spans *v, int32_t id, uint32_t b, uint32_t e, uint32_t nb, uint32_t ne, int lit) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->p = realloc(v->p, sizeof(Span) * v->cap); }
  Span *s = &v->p[v->n++];
  s->id = id; s->begin = b; s->end = e; s->nb = nb; s->ne = ne; s->lit = lit;
}
static int populate_spans(const Oracle *o, const uint8_t *norm, int norm_size, const Align *n2o, uint32_t input_size,
                          const Toks *res, Spans *out) {
  size_t consumed = 0; int is_prev_unk = 0;
  out->n = 0;
  for (size_t k = 0; k < res->n; ++k) {
    const Tok *t = &res->p[k];
    if (t->len == 0) return -1;                             /* :557 */
    const int id = t->id;
    const int type = (id >= 0 && id < o->n_pieces) ? o->pieces[id].type : 0;
    const int is_unk = type == T_UNKNOWN;
    if (type == T_CONTROL) {                                /* :561-567 begin == end */
      if (consumed >= n2o->n) return -1;
      spans_push(out, id, n2o->p[consumed], n2o->p[consumed], (uint32_t)t->begin, (uint32_t)(t->begin + t->len), 0);
    } else {
      const size_t begin = consumed, end = consumed + (size_t)t->len;
      if (begin >= n2o->n || end >= n2o->n) return -1;      /* :570-571 */
      const uint32_t ob = n2o->p[begin], oe = n2o->p[end];
      if (ob > input_size || oe > input_size || ob > oe) return -1;   /* :574-576 */
      if (is_unk && o->byte_fallback) {                     /* :581-603: the last byte piece holds the surface */
        for (int i = 0; i < t->len; ++i)
          spans_push(out, o->byte_ids[norm[t->begin + i]], ob, i == t->len - 1 ? oe : ob, (uint32_t)begin, (uint32_t)end, 1);
      } else if (is_prev_unk && is_unk) {                   /* :609-613 */
        out->p[out->n - 1].end = oe;
        out->p[out->n - 1].ne = (uint32_t)end;
      } else {
        spans_push(out, id, ob, oe, (uint32_t)begin, (uint32_t)end, 0);
      }
      consumed = end;
    }
    is_prev_unk = is_unk;
  }
  if (consumed != (size_t)norm_size) return -1;             /* :628 */
  for (int k = 0; k < o->n_opts; ++k) {                     /* ApplyExtraOptions (:1019-1064) */
    switch (o->opts[k]) {
      case OPT_REVERSE:
        for (size_t i = 0, j = out->n ? out->n - 1 : 0; i < j; ++i, --j) { Span t = out->p[i]; out->p[i] = out->p[j]; out->p[j] = t; }
        break;
      case OPT_EOS: { sv e = o->eos_piece; e.n = strnlen((const char *)e.p, e.n); spans_push(out, piece_to_id(o, e), input_size, input_size, 0, 0, 1); break; }
      case OPT_BOS: {
        sv b = o->bos_piece; b.n = strnlen((const char *)b.p, b.n);
        spans_push(out, 0, 0, 0, 0, 0, 1);
        memmove(out->p + 1, out->p, sizeof(Span) * (out->n - 1));
        out->p[0].id = piece_to_id(o, b); out->p[0].begin = 0; out->p[0].end = 0;
        out->p[0].nb = out->p[0].ne = 0; out->p[0].lit = 1;
        break;
      }
    }
  }
  return 0;
}

/* SentencePieceProcessor::Encode(input, SentencePieceText*) (:638-651) + ids (:392-403) */
static int encode_one(const Oracle *o, const uint8_t *in, size_t n, Buf *norm, Toks *toks, Ids *ids) {
  normalize(o, in, n, norm);
  if (o->model_type == M_UNIGRAM) unigram_encode(o, norm->p, (int)norm->n, toks);
  else bpe_encode(o, norm->p, (int)norm->n, toks);
  return populate_ids(o, norm->p, (int)norm->n, toks, ids);
}

/* ------------------------------------------------------------------ load */
static void seterr(char *err, uint64_t cap, const char *msg) { if (err && cap) { snprintf(err, cap, "%s", msg); } }

static int parse_piece(sv m, Piece *pc) {
  PB b = { m.p, m.p + m.n, 0 }; int wt; uint64_t v; sv s;
  pc->piece.p = (const uint8_t *)""; pc->piece.n = 0; pc->score = 0; pc->type = T_NORMAL;
  int f;
  while ((f = pb_next(&b, &wt, &v, &s))) {
    if (f == 1 && wt == 2) pc->piece = s;
    else if (f == 2 && wt == 5) { uint32_t u = (uint32_t)v; memcpy(&pc->score, &u, 4); }
    else if (f == 3 && wt == 0) pc->type = (int)v;
  }
  return b.err;
}

static sv csv(const char *c) { sv s = { (const uint8_t *)c, strlen(c) }; return s; }

/* Rebuilds everything that depends on piece types: ModelInterface::
 * InitializePieces (model_interface.cc:63-151) and unigram::Model::Model +
 * BuildTrie (unigram_model.cc:652-670, :608-650).  Called at load only; as in
 * the reference, SetVocabulary changes types afterwards WITHOUT rebuilding
 * the maps/trie/min/max (sentencepiece_processor.cc:301-330). */
static int init_model(Oracle *o, char *err, uint64_t errcap) {
  o->unk_id = -1;
  smap_init(&o->pieces_map, (size_t)o->n_pieces);
  smap_init(&o->reserved_map, (size_t)o->n_pieces);
  trie_init(&o->uds); trie_init(&o->ptrie);
  int byte_found[256] = {0};
  for (int i = 0; i < o->n_pieces; ++i) {
    const Piece *sp = &o->pieces[i];
    if (sp->piece.n == 0) { seterr(err, errcap, "piece must not be empty."); return -1; }
    const int is_normal = sp->type == T_NORMAL || sp->type == T_USER_DEFINED || sp->type == T_UNUSED;
    if (!smap_insert(is_normal ? &o->pieces_map : &o->reserved_map, sp->piece, i)) { seterr(err, errcap, "piece is already defined."); return -1; }
    if (sp->type == T_USER_DEFINED) { trie_insert(&o->uds, sp->piece, i); o->has_uds = 1; }
    if (sp->type == T_UNKNOWN) { if (o->unk_id >= 0) { seterr(err, errcap, "unk is already defined."); return -1; } o->unk_id = i; }
    if (sp->type == T_BYTE) {
      if (!o->byte_fallback) { seterr(err, errcap, "byte piece found although byte_fallback is false."); return -1; }
      unsigned b = 256;                                     /* PieceToByte (model_interface.cc:214-229) */
      if (sp->piece.n == 6 && memcmp(sp->piece.p, "<0x", 3) == 0 && sp->piece.p[5] == '>') {
        char hx[3] = { (char)sp->piece.p[3], (char)sp->piece.p[4], 0 }; char chk[8];
        b = (unsigned)strtoul(hx, NULL, 16); snprintf(chk, sizeof chk, "<0x%02X>", b);
        if (memcmp(chk, sp->piece.p, 6) != 0) b = 256;
      }
      if (b > 255) { seterr(err, errcap, "byte piece is invalid."); return -1; }
      byte_found[b] = 1;
    }
  }
  if (o->unk_id == -1) { seterr(err, errcap, "unk is not defined."); return -1; }
  if (o->byte_fallback) for (int b = 0; b < 256; ++b) if (!byte_found[b]) { seterr(err, errcap, "there are not 256 byte pieces although byte_fallback is true."); return -1; }
  /* byte -> id of "<0x%02X>" through PieceToId (sentencepiece_processor.cc:587-588) */
  for (int b = 0; b < 256; ++b) {
    char name[8]; snprintf(name, sizeof name, "<0x%02X>", b);
    o->byte_ids[b] = piece_to_id(o, csv(name));
  }
  if (o->model_type == M_UNIGRAM) {
    o->min_score = FLT_MAX; o->max_score = FLT_MIN;        /* unigram_model.cc:657-664 */
    int any = 0;
    for (int i = 0; i < o->n_pieces; ++i) {
      const Piece *sp = &o->pieces[i];
      if (sp->type == T_NORMAL) { if (sp->score < o->min_score) o->min_score = sp->score; if (sp->score > o->max_score) o->max_score = sp->score; }
      if (sp->type == T_NORMAL || sp->type == T_USER_DEFINED || sp->type == T_UNUSED) { trie_insert(&o->ptrie, sp->piece, i); any = 1; }
    }
    if (!any) { seterr(err, errcap, "no pieces are loaded."); return -1; }
  }
  return 0;
}

Oracle *oracle_load(const void *model_bytes, uint64_t n, char *err, uint64_t errcap) {
  Oracle *o = calloc(1, sizeof(Oracle));
  o->buf = malloc(n ? n : 1); memcpy(o->buf, model_bytes, n); o->buf_n = n;
  o->model_type = M_UNIGRAM; o->add_dummy_prefix = 1; o->remove_extra_ws = 1; o->escape_ws = 1;
  o->unk_piece = csv("<unk>"); o->bos_piece = csv("<s>"); o->eos_piece = csv("</s>"); o->pad_piece = csv("<pad>");
  PB b = { o->buf, o->buf + n, 0 }; int wt, f; uint64_t v; sv s;
  int cap = 0;
  while ((f = pb_next(&b, &wt, &v, &s))) {
    if (f == 1 && wt == 2) {
      if (o->n_pieces == cap) { cap = cap ? cap * 2 : 1024; o->pieces = realloc(o->pieces, sizeof(Piece) * (size_t)cap); }
      if (parse_piece(s, &o->pieces[o->n_pieces++])) b.err = 1;
    } else if (f == 2 && wt == 2) {                          /* trainer_spec */
      PB t = { s.p, s.p + s.n, 0 }; int g; sv ts;
      while ((g = pb_next(&t, &wt, &v, &ts))) {
        if (g == 3 && wt == 0) o->model_type = (int)v;
        else if (g == 24 && wt == 0) o->ws_suffix = v != 0;
        else if (g == 35 && wt == 0) o->byte_fallback = v != 0;
        else if (g == 44 && wt == 2) { o->unk_surface = ts; o->has_unk_surface = 1; }
        else if (g == 45 && wt == 2 && ts.n) o->unk_piece = ts;   /* RETURN_PIECE: empty -> default */
        else if (g == 46 && wt == 2 && ts.n) o->bos_piece = ts;
        else if (g == 47 && wt == 2 && ts.n) o->eos_piece = ts;
        else if (g == 48 && wt == 2 && ts.n) o->pad_piece = ts;
      }
      if (t.err) b.err = 1;
    } else if (f == 3 && wt == 2) {                          /* normalizer_spec */
      PB t = { s.p, s.p + s.n, 0 }; int g; sv ts;
      while ((g = pb_next(&t, &wt, &v, &ts))) {
        if (g == 2 && wt == 2) o->charsmap = ts;
        else if (g == 3 && wt == 0) o->add_dummy_prefix = v != 0;
        else if (g == 4 && wt == 0) o->remove_extra_ws = v != 0;
        else if (g == 5 && wt == 0) o->escape_ws = v != 0;
      }
      if (t.err) b.err = 1;
    } else if (f == 5 && wt == 2) {                          /* denormalizer_spec */
      PB t = { s.p, s.p + s.n, 0 }; int g; sv ts;
      while ((g = pb_next(&t, &wt, &v, &ts))) if (g == 2 && wt == 2 && ts.n) o->has_denormalizer = 1;
      if (t.err) b.err = 1;
    }
  }
  if (b.err) { seterr(err, errcap, "could not parse ModelProto"); oracle_free(o); return NULL; }
  if (o->model_type != M_UNIGRAM && o->model_type != M_BPE) { seterr(err, errcap, "only unigram and bpe are restated"); oracle_free(o); return NULL; }
  /* Normalizer::Init + DecodePrecompiledCharsMap (normalizer.cc:47-69, :274-309) */
  if (o->charsmap.n) {
    uint32_t tsz = 0;
    if (o->charsmap.n <= 4) { seterr(err, errcap, "Blob for normalization rule is broken."); oracle_free(o); return NULL; }
    memcpy(&tsz, o->charsmap.p, 4);
    if (tsz >= o->charsmap.n) { seterr(err, errcap, "Trie data size exceeds the input blob size."); oracle_free(o); return NULL; }
    uint32_t *units = malloc(tsz ? tsz : 4);                 /* aligned copy */
    memcpy(units, o->charsmap.p + 4, tsz);
    o->nunits = units; o->n_nunits = tsz / 4;
    o->nstrings = (const char *)o->charsmap.p + 4 + tsz; o->nstrings_n = o->charsmap.n - 4 - tsz;
  }
  if (init_model(o, err, errcap)) { oracle_free(o); return NULL; }
  return o;
}

void oracle_free(Oracle *o) {
  if (!o) return;
  free(o->buf); free(o->pieces); free((void *)o->nunits);
  free(o->pieces_map.e); free(o->reserved_map.e);
  if (o->uds.first_child) trie_free(&o->uds);
  if (o->ptrie.first_child) trie_free(&o->ptrie);
  free(o);
}

/* ParseExtraOptions (sentencepiece_processor.cc:1067-1101); "unk"/"unk_piece"
 * only changes piece strings and is a no-op for ids. */
int oracle_set_encode_extra_options(Oracle *o, const char *opts) {
  o->n_opts = 0;
  o->opt_unk_piece = 0;
  const char *p = opts;
  while (*p) {
    const char *q = strchr(p, ':'); size_t n = q ? (size_t)(q - p) : strlen(p);
    sv s = { (const uint8_t *)p, n };
    int code;
    if (sv_eq_c(s, "bos")) code = OPT_BOS; else if (sv_eq_c(s, "eos")) code = OPT_EOS;
    else if (sv_eq_c(s, "reverse")) code = OPT_REVERSE; else if (sv_eq_c(s, "unk") || sv_eq_c(s, "unk_piece")) code = -1;
    else return 13;
    if (code == OPT_BOS || code == OPT_EOS) {
      sv pc = code == OPT_BOS ? o->bos_piece : o->eos_piece; pc.n = strnlen((const char *)pc.p, pc.n);
      const int id = piece_to_id(o, pc);
      if (o->pieces[id].type == T_UNKNOWN) return 13;       /* "id for `<s>` is not defined." */
    }
    if (code >= 0 && o->n_opts < 16) o->opts[o->n_opts++] = code;
    if (code < 0) o->opt_unk_piece = 1;
    if (!q) break;
    p = q + 1;
  }
  return 0;
}

/* SetVocabulary / ResetVocabulary (sentencepiece_processor.cc:301-340) */
int oracle_set_vocabulary(Oracle *o, const char *pieces, uint64_t len) {
  SMap vocab; smap_init(&vocab, 1024 + (size_t)len / 2);
  const char *p = pieces, *end = pieces + len;
  while (p < end) {
    const char *q = memchr(p, '\n', (size_t)(end - p)); if (!q) q = end;
    sv s = { (const uint8_t *)p, (size_t)(q - p) }; smap_insert(&vocab, s, 1);
    p = q + 1;
  }
  for (int i = 0; i < o->n_pieces; ++i) {
    Piece *pc = &o->pieces[i];
    if (pc->type == T_CONTROL || pc->type == T_UNKNOWN || pc->type == T_USER_DEFINED) continue;
    if (smap_find(&vocab, pc->piece) >= 0 || (size_t)one_char_len(pc->piece.p[0]) == pc->piece.n) pc->type = T_NORMAL;
    else pc->type = T_UNUSED;
  }
  free(vocab.e);
  return 0;
}
int oracle_reset_vocabulary(Oracle *o) {
  for (int i = 0; i < o->n_pieces; ++i) if (o->pieces[i].type == T_UNUSED) o->pieces[i].type = T_NORMAL;
  return 0;
}

int64_t oracle_normalize(const Oracle *o, const char *in, uint64_t n, char *out, uint64_t cap) {
  Buf b = {0};
  normalize(o, (const uint8_t *)in, (size_t)n, &b);
  int64_t r = (int64_t)b.n;
  if (b.n > cap) r = -(int64_t)b.n - 2; else if (b.n) memcpy(out, b.p, b.n);
  free(b.p);
  return r;
}

/* Normalize(input, &normalized, &norm_to_orig) per sentence.  n2o (optional): sentence s owns entries
 * [norm_offsets[s] + s, norm_offsets[s + 1] + s + 1); 0xFFFFFFFF stands for "the reference's vector is empty". */
int64_t oracle_normalize_batch(const Oracle *o, const char *text, const uint64_t *offsets, uint64_t n,
                               char *out, uint64_t cap, uint64_t *norm_offsets, uint32_t *n2o) {
  Buf b = {0}; Align a = {0};
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    norm_offsets[i] = total;
    normalize_aligned(o, (const uint8_t *)text + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), &b, &a);
    if (total + b.n <= cap) {
      if (b.n) memcpy(out + total, b.p, b.n);
      if (n2o) {
        if (a.n == 0) n2o[total + i] = 0xFFFFFFFFu;
        else memcpy(n2o + total + i, a.p, sizeof(uint32_t) * a.n);
      }
    }
    total += b.n;
  }
  norm_offsets[n] = total;
  free(b.p); free(a.p);
  if (total > cap) return -(int64_t)total - 2;
  return (int64_t)total;
}

int64_t oracle_encode(const Oracle *o, const char *in, uint64_t n, int32_t *out, uint64_t cap) {
  Buf norm = {0}; Toks toks = {0}; Ids ids = {0};
  int64_t r;
  if (encode_one(o, (const uint8_t *)in, (size_t)n, &norm, &toks, &ids)) r = -1;
  else if (ids.n > cap) r = -(int64_t)ids.n - 2;
  else { if (ids.n) memcpy(out, ids.p, sizeof(int32_t) * ids.n); r = (int64_t)ids.n; }
  free(norm.p); free(toks.p); free(ids.p);
  return r;
}

int64_t oracle_encode_batch(const Oracle *o, const char *text, const uint64_t *offsets, uint64_t n,
                            int32_t *out, uint64_t cap, uint64_t *id_offsets) {
  Buf norm = {0}; Toks toks = {0}; Ids ids = {0};
  uint64_t total = 0; int failed = 0;
  for (uint64_t i = 0; i < n; ++i) {
    id_offsets[i] = total;
    if (encode_one(o, (const uint8_t *)text + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), &norm, &toks, &ids)) { failed = 1; break; }
    if (total + ids.n <= cap && ids.n) memcpy(out + total, ids.p, sizeof(int32_t) * ids.n);
    total += ids.n;
  }
  id_offsets[n] = total;
  free(norm.p); free(toks.p); free(ids.p);
  if (failed) return -1;
  if (total > cap) return -(int64_t)total - 2;
  return (int64_t)total;
}

/* Encode(input, SentencePieceText*) per sentence: ids + pieces(i).begin / .end (bytes, relative to the sentence). */
int64_t oracle_encode_pieces_batch(const Oracle *o, const char *text, const uint64_t *offsets, uint64_t n,
                                   int32_t *out, uint32_t *begin, uint32_t *end, uint64_t cap, uint64_t *id_offsets,
                                   char *pieces, uint64_t pieces_cap, uint64_t *piece_offsets);
int64_t oracle_encode_spans_batch(const Oracle *o, const char *text, const uint64_t *offsets, uint64_t n,
                                  int32_t *out, uint32_t *begin, uint32_t *end, uint64_t cap, uint64_t *id_offsets) {
  return oracle_encode_pieces_batch(o, text, offsets, n, out, begin, end, cap, id_offsets, NULL, 0, NULL);
}

/* The same with pieces(i).piece(): packed into `pieces`, piece k of the batch at [piece_offsets[k], piece_offsets[k + 1])
 * (piece_offsets holds cap + 1 entries).  Returns -3 - (bytes needed) if pieces_cap is too small. */
int64_t oracle_encode_pieces_batch(const Oracle *o, const char *text, const uint64_t *offsets, uint64_t n,
                                   int32_t *out, uint32_t *begin, uint32_t *end, uint64_t cap, uint64_t *id_offsets,
                                   char *pieces, uint64_t pieces_cap, uint64_t *piece_offsets) {
  Buf norm = {0}; Toks toks = {0}; Spans sp = {0}; Align n2o = {0};
  uint64_t total = 0, pbytes = 0; int failed = 0;
  for (uint64_t i = 0; i < n; ++i) {
    id_offsets[i] = total;
    const uint8_t *in = (const uint8_t *)text + offsets[i];
    const size_t len = (size_t)(offsets[i + 1] - offsets[i]);
    normalize_aligned(o, in, len, &norm, &n2o);
    if (o->model_type == M_UNIGRAM) unigram_encode(o, norm.p, (int)norm.n, &toks);
    else bpe_encode(o, norm.p, (int)norm.n, &toks);
    if (populate_spans(o, norm.p, (int)norm.n, &n2o, (uint32_t)len, &toks, &sp)) { failed = 1; break; }
    if (total + sp.n <= cap)
      for (size_t k = 0; k < sp.n; ++k) {
        const Span *x = &sp.p[k];
        out[total + k] = x->id; begin[total + k] = x->begin; end[total + k] = x->end;
        if (piece_offsets) {
          /* :563 / :592 / :616 the normalized text; ByteToPiece (:590); bos / eos names (:1033, :1045); UNK_PIECE (:1050-1058) */
          const int unk = x->id >= 0 && x->id < o->n_pieces && o->pieces[x->id].type == T_UNKNOWN;
          sv ps; ps.p = norm.p + x->nb; ps.n = x->ne - x->nb;
          if (x->lit || (unk && o->opt_unk_piece)) ps = o->pieces[x->id].piece;
          piece_offsets[total + k] = pbytes;
          if (pbytes + ps.n <= pieces_cap && ps.n) memcpy(pieces + pbytes, ps.p, ps.n);
          pbytes += ps.n;
        }
      }
    total += sp.n;
  }
  id_offsets[n] = total;
  if (piece_offsets && total <= cap) piece_offsets[total] = pbytes;
  free(norm.p); free(toks.p); free(sp.p); free(n2o.p);
  if (failed) return -1;
  if (total > cap) return -(int64_t)total - 2;
  if (pbytes > pieces_cap) return -(int64_t)pbytes - 3;
  return (int64_t)total;
}

/* ------------------------------------------------------------------ n-best */
/* unigram::Model::NBestEncode (unigram_model.cc:686-717): Lattice::SetSentence (:114-152), Model::PopulateNodes
 * (:547-596), Lattice::Viterbi (:167-198, all float, first best wins ties) and the A* of Lattice::NBest (:345-515).
 * The agenda is std::priority_queue over fx: its order among equal keys is the order libstdc++'s push_heap /
 * pop_heap produce, so those two are restated step by step (bits/stl_heap.h __push_heap / __adjust_heap). */
typedef struct { int pos, length, id, byte_begin, byte_len; float score, backtrace_score; int prev; } LNode;
typedef struct { LNode *p; size_t n, cap; } LNodes;
typedef struct { int *p; size_t n, cap; } IVec;
static void ivec_push(IVec *v, int x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 16; v->p = realloc(v->p, sizeof(int) * v->cap); }
  v->p[v->n++] = x;
}
static int lnode_new(LNodes *v) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->p = realloc(v->p, sizeof(LNode) * v->cap); }
  memset(&v->p[v->n], 0, sizeof(LNode));            /* FreeList hands out zeroed nodes */
  v->p[v->n].prev = -1;
  return (int)v->n++;
}
typedef struct { int node, next; float fx, gx; } Hyp;
typedef struct { Hyp *p; size_t n, cap; } Hyps;
static int hyp_new(Hyps *v) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 512; v->p = realloc(v->p, sizeof(Hyp) * v->cap); }
  return (int)v->n++;
}
/* std::push_heap / std::pop_heap with comp(a, b) = a->fx < b->fx on an array of hypothesis indices */
static void heap_push_fx(IVec *h, const Hyps *hy, int value) {
  ivec_push(h, value);
  long hole = (long)h->n - 1, parent = (hole - 1) / 2;
  while (hole > 0 && hy->p[h->p[parent]].fx < hy->p[value].fx) {
    h->p[hole] = h->p[parent]; hole = parent; parent = (hole - 1) / 2;
  }
  h->p[hole] = value;
}
static int heap_pop_fx(IVec *h, const Hyps *hy) {
  const int top = h->p[0];
  const int value = h->p[h->n - 1];
  const long len = (long)h->n - 1;                  /* the range the hole is sifted in */
  h->n--;
  if (len == 0) return top;
  long hole = 0, child = 0;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (hy->p[h->p[child]].fx < hy->p[h->p[child - 1]].fx) child--;
    h->p[hole] = h->p[child]; hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    h->p[hole] = h->p[child - 1]; hole = child - 1;
  }
  long parent = (hole - 1) / 2;
  while (hole > 0 && hy->p[h->p[parent]].fx < hy->p[value].fx) {
    h->p[hole] = h->p[parent]; hole = parent; parent = (hole - 1) / 2;
  }
  h->p[hole] = value;
  return top;
}

typedef struct { Toks toks; float score; } NBestOne;
/* Fills results (caller frees each toks.p); returns the number of results. */
static int unigram_nbest(const Oracle *o, const uint8_t *norm, int size, int nbest_size, NBestOne *results) {
  /* SetSentence: character starts */
  IVec surf = {0};
  for (int p = 0; p < size;) { ivec_push(&surf, p); int mb = one_char_len(norm[p]); if (mb > size - p) mb = size - p; p += mb; }
  ivec_push(&surf, size);
  const int len = (int)surf.n - 1;
  LNodes nodes = {0};
  IVec *begin_nodes = calloc((size_t)len + 1, sizeof(IVec)), *end_nodes = calloc((size_t)len + 1, sizeof(IVec));
  const int bos = lnode_new(&nodes); nodes.p[bos].id = -1; nodes.p[bos].pos = 0; ivec_push(&end_nodes[0], bos);
  const int eos = lnode_new(&nodes); nodes.p[eos].id = -1; nodes.p[eos].pos = len; ivec_push(&begin_nodes[len], eos);
  /* PopulateNodes */
  const float unk_score = o->min_score - 10.0f;
  for (int begin_pos = 0; begin_pos < len; ++begin_pos) {
    const int b0 = surf.p[begin_pos];
    int has_single_node = 0, node = 0;
    for (int key_pos = b0; key_pos < size;) {                 /* commonPrefixSearch: matches in increasing length */
      node = trie_child(&o->ptrie, node, norm[key_pos]);
      if (node < 0) break;
      ++key_pos;
      const int id = o->ptrie.value[node];
      if (id < 0) continue;
      int pos = begin_pos;                                    /* get_chars_length (:548-552) */
      while (surf.p[pos] < key_pos) ++pos;
      const int length = pos - begin_pos;
      if (o->pieces[id].type == T_UNUSED) continue;
      const int nd = lnode_new(&nodes);
      LNode *x = &nodes.p[nd];
      x->pos = begin_pos; x->length = length; x->byte_begin = b0; x->byte_len = surf.p[begin_pos + length] - b0;
      x->id = id;
      x->score = o->pieces[id].type == T_USER_DEFINED ? (float)((double)((float)length * o->max_score) - 0.1)
                                                       : o->pieces[id].score;
      ivec_push(&begin_nodes[begin_pos], nd); ivec_push(&end_nodes[begin_pos + length], nd);
      if (!has_single_node && length == 1) has_single_node = 1;
    }
    if (!has_single_node) {
      const int nd = lnode_new(&nodes);
      LNode *x = &nodes.p[nd];
      x->pos = begin_pos; x->length = 1; x->byte_begin = b0; x->byte_len = surf.p[begin_pos + 1] - b0;
      x->id = o->unk_id; x->score = unk_score;
      ivec_push(&begin_nodes[begin_pos], nd); ivec_push(&end_nodes[begin_pos + 1], nd);
    }
  }
  /* Viterbi: fills backtrace_score = h(node) */
  for (int pos = 0; pos <= len; ++pos) {
    for (size_t r = 0; r < begin_nodes[pos].n; ++r) {
      LNode *rn = &nodes.p[begin_nodes[pos].p[r]];
      float best_score = 0.0f; int best_node = -1;
      for (size_t l = 0; l < end_nodes[pos].n; ++l) {
        const int ln = end_nodes[pos].p[l];
        const float sc = nodes.p[ln].backtrace_score + rn->score;
        if (best_node < 0 || sc > best_score) { best_node = ln; best_score = sc; }
      }
      rn->prev = best_node; rn->backtrace_score = best_score;
    }
  }
  /* A* from EOS */
  Hyps hy = {0}; IVec agenda = {0};
  int n_res = 0;
  { const int e = hyp_new(&hy); hy.p[e].node = eos; hy.p[e].next = -1; hy.p[e].gx = 0.0f; hy.p[e].fx = nodes.p[eos].backtrace_score;
    heap_push_fx(&agenda, &hy, e); }
  while (agenda.n) {
    const int top = heap_pop_fx(&agenda, &hy);
    const int node = hy.p[top].node;
    if (node == bos) {
      NBestOne *r = &results[n_res++];
      memset(r, 0, sizeof(*r));
      for (int h = hy.p[top].next; hy.p[h].next != -1; h = hy.p[h].next) {
        const LNode *x = &nodes.p[hy.p[h].node];
        toks_push(&r->toks, x->byte_begin, x->byte_len, x->id);
      }
      r->score = hy.p[top].fx;
      if (n_res == nbest_size) break;
      continue;
    }
    const IVec *en = &end_nodes[nodes.p[node].pos];
    for (size_t i = 0; i < en->n; ++i) {
      const LNode *ln = &nodes.p[en->p[i]];
      const int h = hyp_new(&hy);
      hy.p[h].node = en->p[i];
      hy.p[h].gx = ln->score + hy.p[top].gx;
      hy.p[h].fx = ln->backtrace_score + hy.p[top].gx;
      hy.p[h].next = top;
      heap_push_fx(&agenda, &hy, h);
    }
    if (agenda.n >= 10000) {                                  /* :487-514 keep the best min(512, 10 nbest) */
      int keep = nbest_size * 10 < 512 ? nbest_size * 10 : 512;
      IVec na = {0};
      for (int i = 0; i < keep; ++i) heap_push_fx(&na, &hy, heap_pop_fx(&agenda, &hy));
      free(agenda.p); agenda = na;
    }
  }
  for (int i = 0; i <= len; ++i) { free(begin_nodes[i].p); free(end_nodes[i].p); }
  free(begin_nodes); free(end_nodes); free(nodes.p); free(hy.p); free(agenda.p); free(surf.p);
  return n_res;
}

/* SentencePieceProcessor::NBestEncode(input, nbest_size, std::vector<std::vector<int>>*)
 * (sentencepiece_processor.cc:451-467, :653-678).  ids of result k at out[offs[k], offs[k + 1]); scores[k].
 * Returns the number of results, -1 on an error status (not a unigram model, ...), -(needed) - 2 if cap is small. */
int64_t oracle_nbest_encode(const Oracle *o, const char *in, uint64_t n, int nbest_size, int32_t *out, uint64_t cap,
                            uint64_t *offs, float *scores) {
  if (o->model_type != M_UNIGRAM) return -1;                  /* IsNBestEncodeAvailable (:662) */
  Buf norm = {0}; Ids ids = {0};
  normalize(o, (const uint8_t *)in, (size_t)n, &norm);
  if (nbest_size > 1024) nbest_size = 1024;                   /* unigram_model.cc:692 */
  if (nbest_size < 1) nbest_size = 1;
  NBestOne *res = calloc((size_t)nbest_size, sizeof(NBestOne));
  int n_res;
  if (norm.n == 0) { n_res = 1; res[0].score = 0.0f; }        /* :688-690 one empty result */
  else if (nbest_size <= 1) { n_res = 1; unigram_encode(o, norm.p, (int)norm.n, &res[0].toks); res[0].score = 0.0f; }   /* :694-696 */
  else n_res = unigram_nbest(o, norm.p, (int)norm.n, nbest_size, res);
  uint64_t total = 0; int failed = n_res == 0;                /* "NBestEncode returns empty result." (:666) */
  for (int k = 0; k < n_res && !failed; ++k) {
    offs[k] = total;
    if (populate_ids(o, norm.p, (int)norm.n, &res[k].toks, &ids)) { failed = 1; break; }
    if (total + ids.n <= cap && ids.n) memcpy(out + total, ids.p, sizeof(int32_t) * ids.n);
    total += ids.n;
    scores[k] = res[k].score;
  }
  offs[n_res] = total;
  for (int k = 0; k < nbest_size; ++k) free(res[k].toks.p);
  free(res); free(norm.p); free(ids.p);
  if (failed) return -1;
  if (total > cap) return -(int64_t)total - 2;
  return n_res;
}

/* ------------------------------------------------------------------ decode */
/* SentencePieceProcessor::Decode(const std::vector<int>& ids, std::string*)
 * (sentencepiece_processor.cc:761-925): ids -> IdToPiece -> the piece-level Decode, text only.
 * Returns 0, 11 (OUT_OF_RANGE "Invalid id"), 12 (a denormalizer is not restated). */
static const uint8_t kSpaceSym[3] = { 0xE2, 0x96, 0x81 };
static void buf_put(Buf *b, const uint8_t *p, size_t n) {
  if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 64; b->p = realloc(b->p, b->cap); }
  memcpy(b->p + b->n, p, n); b->n += n;
}
/* ProcessBytePieces (:825-880): one Unicode character at a time; a structurally invalid byte -> U+FFFD */
static void flush_bytes(Buf *bytes, Buf *text) {
  size_t off = 0;
  while (off < bytes->n) {
    size_t consumed;
    if (!is_valid_decode_utf8(bytes->p + off, bytes->n - off, &consumed)) {
      static const uint8_t rep[3] = { 0xEF, 0xBF, 0xBD };
      buf_put(text, rep, 3);                               /* consumed == 1 */
    } else {
      buf_put(text, bytes->p + off, consumed);
    }
    off += consumed;
  }
  bytes->n = 0;
}
static int decode_ids(const Oracle *o, const int32_t *ids, size_t n, Buf *text) {
  static const uint8_t kDefaultUnk[] = { ' ', 0xE2, 0x81, 0x87, ' ' };   /* kDefaultUnknownSymbol (:52) */
  sv unk_surface = { kDefaultUnk, 5 };
  if (o->has_unk_surface) {                                /* .c_str(): up to the first NUL (:772-773) */
    unk_surface = o->unk_surface;
    const void *z = memchr(unk_surface.p, 0, unk_surface.n);
    if (z) unk_surface.n = (size_t)((const uint8_t *)z - unk_surface.p);
  }
  if (o->has_denormalizer) return 12;
  for (size_t i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= o->n_pieces) return 11;   /* :913-917 */
  Buf bytes = { 0, 0, 0 };
  const size_t text0 = text->n;
  int is_bos_ws = 1, bos_ws_seen = 0;
  for (size_t i = 0; i < n; ++i) {
    const sv piece = o->pieces[ids[i]].piece;
    const int id = piece_to_id(o, piece);                  /* sp->set_id(PieceToId(w)) (:812) */
    const int type = o->pieces[id].type;
    if (type == T_BYTE) {                                  /* PieceToByte("<0xHH>") (:836) */
      unsigned v = 0;
      for (int k = 3; k < 5; ++k) {
        const uint8_t c = piece.p[k];
        v = v * 16 + (c <= '9' ? c - '0' : (c | 0x20) - 'a' + 10);
      }
      const uint8_t bb = (uint8_t)v;
      buf_put(&bytes, &bb, 1);
      continue;
    }
    flush_bytes(&bytes, text);
    if (bos_ws_seen || text->n > text0) is_bos_ws = 0;     /* :893 */
    if (type == T_CONTROL) { bos_ws_seen = 0; continue; }  /* :780-781 invisible */
    if (type == T_UNKNOWN) {                               /* :782-788 */
      if (sv_eq(o->pieces[id].piece, piece)) buf_put(text, unk_surface.p, unk_surface.n);
      else buf_put(text, piece.p, piece.n);
      bos_ws_seen = 0;
      continue;
    }
    const uint8_t *p = piece.p; size_t m = piece.n;
    int has_bos_ws = 0;
    if (is_bos_ws && (o->add_dummy_prefix || o->remove_extra_ws)) {   /* :791-805 */
      if (m >= 3 && memcmp(p, kSpaceSym, 3) == 0) { p += 3; m -= 3; has_bos_ws = 1; }
      if (o->remove_extra_ws) has_bos_ws = 0;
    }
    for (size_t k = 0; k < m;) {                           /* StrReplaceAll(piece, {{kSpaceSymbol, " "}}) (:807) */
      if (k + 3 <= m && memcmp(p + k, kSpaceSym, 3) == 0) { const uint8_t sp = ' '; buf_put(text, &sp, 1); k += 3; }
      else { buf_put(text, p + k, 1); ++k; }
    }
    bos_ws_seen = has_bos_ws;
  }
  flush_bytes(&bytes, text);
  free(bytes.p);
  return 0;
}

int64_t oracle_decode_batch(const Oracle *o, const int32_t *ids, const uint64_t *id_offsets, uint64_t n,
                            char *text, uint64_t cap, uint64_t *text_offsets) {
  Buf b = { 0, 0, 0 };
  for (uint64_t i = 0; i < n; ++i) {
    text_offsets[i] = b.n;
    const int rc = decode_ids(o, ids + id_offsets[i], (size_t)(id_offsets[i + 1] - id_offsets[i]), &b);
    if (rc) { free(b.p); return -rc * 1000; }
  }
  text_offsets[n] = b.n;
  const int64_t total = (int64_t)b.n;
  if ((uint64_t)total > cap) { free(b.p); return -total - 2; }
  if (total) memcpy(text, b.p, (size_t)total);
  free(b.p);
  return total;
}

int oracle_piece_size(const Oracle *o) { return o->n_pieces; }
int oracle_model_type(const Oracle *o) { return o->model_type; }
