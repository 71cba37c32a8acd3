// t1k_amd/csrc/host/inflate.cpp -- a gzip (RFC 1952) / DEFLATE (RFC 1951) decoder that says how far it has got.
//
// The reference reads its input through zlib's gzread, record by record (ReadFiles.hpp:13, 95; kseq.h:94-150): the first read is in hand a
// few microseconds after the file is opened.  Round 4 inflated an ordinary .gz whole with libdeflate before the first read was looked at
// (2.2 s for 3.2 GB of text, in front of a 2.6 s device loop) because libdeflate has no streaming interface and zlib's is four times
// slower.  This decoder is one call over the whole mapped file like libdeflate's -- input and output are both entirely in memory, so there
// is nothing to suspend and resume -- but it PUBLISHES the number of bytes that are final after every DEFLATE block, so that the record
// indexer (host/reads.cpp) and the device loop behind it run on the text while the rest is still being inflated.
//
// Decoding: canonical Huffman codes through one primary table (11 bits literal/length, 9 bits distance) with sub-tables for longer
// codes, a 64-bit bit buffer refilled with one unaligned load, up to three literals per refill, matches copied eight bytes at a time
// (byte-wise when they overlap closer than that).  Plain C++, no device code; checked against zlib on the CPU (tests/test_inflate_cpu.py).
#include <atomic>
#include <cstdint>
#include <cstring>
#include <string>
#include "t1k_host.h"

namespace t1k {
namespace {

enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
// entry: bits 0..7 code bits to consume | 8..11 kind | 12..15 extra bits (K_LEN / distance) or sub-table index bits (K_SUB) | 16..31 value
static inline uint32_t mk(uint32_t nbits, uint32_t kind, uint32_t aux, uint32_t value) { return nbits | (kind << 8) | (aux << 12) | (value << 16); }

constexpr int LBITS = 11, DBITS = 9;
constexpr int LSIZE = (1 << LBITS) + 288 * 16, DSIZE = (1 << DBITS) + 32 * 64;

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t rev(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
  return r;
}

// kind of table: 0 literal/length, 1 distance, 2 code lengths.  Returns false for an over-subscribed code; an incomplete one leaves K_BAD
// entries (legal only for a distance code with a single symbol, RFC 1951 3.2.7: the entries that are there decode as usual).
static bool buildTable(const uint8_t *lens, int n, int tbits, int what, uint32_t *tab, int tabSize) {
  int count[16] = {0};
  for (int i = 0; i < n; ++i) ++count[lens[i]];
  count[0] = 0;
  uint32_t next[16];
  uint32_t code = 0;
  int64_t left = 1;
  for (int l = 1; l <= 15; ++l) {
    left = (left << 1) - count[l];
    if (left < 0) return false;
    code = (code + count[l - 1]) << 1;
    next[l] = code;
  }
  const int P = 1 << tbits;
  for (int i = 0; i < P; ++i) tab[i] = mk(0, K_BAD, 0, 0);
  auto entryOf = [&](int sym, uint32_t nbits) -> uint32_t {
    if (what == 0) {
      if (sym < 256) return mk(nbits, K_LIT, 0, (uint32_t)sym);
      if (sym == 256) return mk(nbits, K_EOB, 0, 0);
      if (sym > 285) return mk(nbits, K_BAD, 0, 0);
      return mk(nbits, K_LEN, kLenExtra[sym - 257], kLenBase[sym - 257]);
    }
    if (what == 1) {
      if (sym > 29) return mk(nbits, K_BAD, 0, 0);
      return mk(nbits, K_LEN, kDistExtra[sym], kDistBase[sym]);
    }
    return mk(nbits, K_LIT, 0, (uint32_t)sym);
  };
  // codes that fit the primary table; the longest code under every primary index that needs a sub-table
  uint8_t subMax[1 << LBITS];
  memset(subMax, 0, (size_t)P);
  struct Long { uint32_t r; uint8_t len; uint16_t sym; };
  static thread_local Long longs[288];
  int nLong = 0;
  for (int sym = 0; sym < n; ++sym) {
    const int l = lens[sym];
    if (!l) continue;
    const uint32_t r = rev(next[l]++, l);
    if (l <= tbits) {
      const uint32_t e = entryOf(sym, (uint32_t)l);
      for (uint32_t i = r; i < (uint32_t)P; i += 1u << l) tab[i] = e;
    } else {
      const uint32_t p = r & (P - 1);
      if (l > subMax[p]) subMax[p] = (uint8_t)l;
      longs[nLong++] = {r, (uint8_t)l, (uint16_t)sym};
    }
  }
  int used = P;
  for (int p = 0; p < P; ++p)
    if (subMax[p]) {
      const int sb = subMax[p] - tbits;
      if (used + (1 << sb) > tabSize) return false;
      tab[p] = mk((uint32_t)tbits, K_SUB, (uint32_t)sb, (uint32_t)used);
      for (int i = 0; i < (1 << sb); ++i) tab[used + i] = mk(0, K_BAD, 0, 0);
      used += 1 << sb;
    }
  for (int i = 0; i < nLong; ++i) {
    const Long &L = longs[i];
    const uint32_t p = L.r & (P - 1), e0 = tab[p];
    const int sb = (int)((e0 >> 12) & 15), rest = L.len - tbits;
    const uint32_t base = e0 >> 16, e = entryOf(L.sym, (uint32_t)rest);
    for (uint32_t j = L.r >> tbits; j < (1u << sb); j += 1u << rest) tab[base + j] = e;
  }
  return true;
}

struct Bits {
  const uint8_t *in, *end;
  uint64_t buf = 0;
  int cnt = 0;
  bool under = false;  // more bits were used than the stream holds (seen at the next refill: every decoding step starts with one)
  inline void refill() {
    if (cnt < 0) { under = true; cnt = 0; buf = 0; }
    if (end - in >= 8) {
      uint64_t w;
      memcpy(&w, in, 8);
      buf |= w << cnt;
      in += (63 - cnt) >> 3;
      cnt |= 56;
    } else {
      while (cnt <= 56 && in < end) { buf |= (uint64_t)*in++ << cnt; cnt += 8; }
    }
  }
  inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
  inline void drop(int n) { buf >>= n; cnt -= n; }
  inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

static const uint32_t *fixedLit() {
  static uint32_t tab[LSIZE];
  static const bool once = [] {
    uint8_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    return buildTable(l, 288, LBITS, 0, tab, LSIZE);
  }();
  (void)once;
  return tab;
}
static const uint32_t *fixedDist() {
  static uint32_t tab[DSIZE];
  static const bool once = [] {
    uint8_t l[32];
    for (int i = 0; i < 32; ++i) l[i] = 5;
    return buildTable(l, 32, DBITS, 1, tab, DSIZE);
  }();
  (void)once;
  return tab;
}

// one DEFLATE stream: out .. outEnd, `base` = first byte a distance may reach back to.  Returns nullptr on error (msg set).
static uint8_t *inflateRaw(Bits &b, uint8_t *base, uint8_t *out, uint8_t *outEnd, GzProgress *pg, uint8_t *dst0, uint64_t progressBase, const char *&msg) {
  static thread_local uint32_t litTab[LSIZE], distTab[DSIZE], clTab[1 << 7];
  // the output range is full in the middle of a block: what was decoded so far is text, and is published as such (the streaming reader's
  // eligibility check decodes the head of a file into a small range and looks at what arrived)
  auto fullAt = [&](const uint8_t *o) { if (pg) pg->produced.store(progressBase + (uint64_t)(o - dst0), std::memory_order_release); };
  for (;;) {
    b.refill();
    if (b.cnt < 3) { msg = "truncated stream"; return nullptr; }
    const uint32_t bfinal = b.take(1), btype = b.take(2);
    const uint32_t *lit, *dist;
    if (btype == 0) {
      b.drop(b.cnt & 7);
      // the bit buffer holds whole bytes now: hand them back
      b.in -= b.cnt >> 3; b.buf = 0; b.cnt = 0;
      if (b.end - b.in < 4) { msg = "truncated stored block"; return nullptr; }
      const uint32_t len = b.in[0] | (b.in[1] << 8), nlen = b.in[2] | (b.in[3] << 8);
      b.in += 4;
      if ((len ^ 0xFFFFu) != nlen) { msg = "stored block: length check failed"; return nullptr; }
      if ((size_t)(b.end - b.in) < len) { msg = "truncated stored block"; return nullptr; }
      if ((size_t)(outEnd - out) < len) { msg = "more text than the output range holds"; fullAt(out); return nullptr; }
      memcpy(out, b.in, len);
      out += len; b.in += len;
    } else if (btype == 3) {
      msg = "reserved block type";
      return nullptr;
    } else {
      if (btype == 1) { lit = fixedLit(); dist = fixedDist(); }
      else {
        b.refill();
        const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
        if (hlit > 286 || hdist > 30) { msg = "too many length or distance codes"; return nullptr; }
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; ++i) { if (b.cnt < 3) b.refill(); cl[order[i]] = (uint8_t)b.take(3); }
        if (!buildTable(cl, 19, 7, 2, clTab, 1 << 7)) { msg = "bad code-length code"; return nullptr; }
        uint8_t lens[286 + 30 + 138] = {0};
        int n = 0;
        while (n < hlit + hdist) {
          b.refill();
          if (b.under) { msg = "truncated stream"; return nullptr; }
          const uint32_t e = clTab[b.peek(7)];
          if (((e >> 8) & 15) != K_LIT || (int)(e & 255) > b.cnt) { msg = "bad code-length symbol"; return nullptr; }
          b.drop((int)(e & 255));
          const int sym = (int)(e >> 16);
          if (sym < 16) lens[n++] = (uint8_t)sym;
          else {
            int rep, val = 0;
            if (sym == 16) { if (!n) { msg = "repeat without a length before it"; return nullptr; } val = lens[n - 1]; rep = 3 + (int)b.take(2); }
            else if (sym == 17) rep = 3 + (int)b.take(3);
            else rep = 11 + (int)b.take(7);
            if (n + rep > hlit + hdist) { msg = "code lengths run over"; return nullptr; }
            while (rep--) lens[n++] = (uint8_t)val;
          }
        }
        if (!lens[256]) { msg = "no end-of-block code"; return nullptr; }
        if (!buildTable(lens, hlit, LBITS, 0, litTab, LSIZE) || !buildTable(lens + hlit, hdist, DBITS, 1, distTab, DSIZE)) { msg = "over-subscribed code"; return nullptr; }
        lit = litTab; dist = distTab;
      }
      // ---- symbols ----
      // fast loop: while 16 bytes of input and 330 bytes of output room are left nothing below can run out of either (one refill serves
      // three literals or a length + distance pair: 48 bits at most; a match writes at most 258 + 7 bytes), so it carries no checks;
      // it leaves the symbol it cannot finish to the careful loop behind it
      {
        const uint8_t *inStop = b.end - 16;
        uint8_t *outStop = outEnd - 330;
        bool eob = false;
        if (b.cnt < 0) { msg = "truncated stream"; return nullptr; }
        while (b.in <= inStop && out <= outStop) {
          { uint64_t w; memcpy(&w, b.in, 8); b.buf |= w << b.cnt; b.in += (63 - b.cnt) >> 3; b.cnt |= 56; }
          uint32_t e = lit[b.buf & ((1u << LBITS) - 1)];
          if (((e >> 8) & 15) == K_SUB) { b.drop(LBITS); e = lit[(e >> 16) + b.peek((int)((e >> 12) & 15))]; }
          const uint32_t kind = (e >> 8) & 15;
          b.drop((int)(e & 255));
          if (kind == K_LIT) {
            *out++ = (uint8_t)(e >> 16);
            e = lit[b.buf & ((1u << LBITS) - 1)];
            if (((e >> 8) & 15) != K_LIT) continue;
            b.drop((int)(e & 255)); *out++ = (uint8_t)(e >> 16);
            e = lit[b.buf & ((1u << LBITS) - 1)];
            if (((e >> 8) & 15) != K_LIT) continue;
            b.drop((int)(e & 255)); *out++ = (uint8_t)(e >> 16);
            continue;
          }
          if (kind != K_LEN) {
            if (kind == K_EOB) { eob = true; break; }
            msg = "invalid literal/length code";
            return nullptr;
          }
          const uint32_t len = (e >> 16) + b.take((int)((e >> 12) & 15));
          uint32_t d = dist[b.buf & ((1u << DBITS) - 1)];
          if (((d >> 8) & 15) == K_SUB) { b.drop(DBITS); d = dist[(d >> 16) + b.peek((int)((d >> 12) & 15))]; }
          if (((d >> 8) & 15) != K_LEN) { msg = "invalid distance code"; return nullptr; }
          b.drop((int)(d & 255));
          const uint32_t distance = (d >> 16) + b.take((int)((d >> 12) & 15));
          if ((size_t)(out - base) < distance) { msg = "distance reaches before the stream's start"; return nullptr; }
          const uint8_t *f = out - distance;
          uint8_t *o = out, *stop = out + len;
          if (distance >= 16) {
            do { uint64_t w0, w1; memcpy(&w0, f, 8); memcpy(&w1, f + 8, 8); memcpy(o, &w0, 8); memcpy(o + 8, &w1, 8); o += 16; f += 16; } while (o < stop);
          } else if (distance >= 8) {
            do { uint64_t w; memcpy(&w, f, 8); memcpy(o, &w, 8); o += 8; f += 8; } while (o < stop);
          } else if (distance == 1) {
            memset(o, *f, len);
          } else {
            // a pattern of `distance` bytes repeated: build eight bytes of it, then store them at multiples of the distance that is >= 8 apart
            do { *o++ = *f++; } while (o < stop);
          }
          out = stop;
        }
        if (eob) goto blockDone;
      }
      for (;;) {
        b.refill();
        if (b.under) { msg = "truncated stream"; return nullptr; }
        uint32_t e = lit[b.peek(LBITS)];
        if (((e >> 8) & 15) == K_SUB) { b.drop(LBITS); e = lit[(e >> 16) + b.peek((int)((e >> 12) & 15))]; }
        uint32_t kind = (e >> 8) & 15;
        if (kind == K_LIT) {
          // up to three literals per refill (3 x 15 bits <= 56)
          if (out + 3 > outEnd) {
            if (out >= outEnd) { msg = "more text than the output range holds"; fullAt(out); return nullptr; }
            b.drop((int)(e & 255)); *out++ = (uint8_t)(e >> 16);
            continue;
          }
          b.drop((int)(e & 255)); *out++ = (uint8_t)(e >> 16);
          e = lit[b.peek(LBITS)];
          if (((e >> 8) & 15) != K_LIT) continue;   // (sub-table, length, end of block: the loop's top handles it after its refill)
          b.drop((int)(e & 255)); *out++ = (uint8_t)(e >> 16);
          e = lit[b.peek(LBITS)];
          if (((e >> 8) & 15) != K_LIT) continue;
          b.drop((int)(e & 255)); *out++ = (uint8_t)(e >> 16);
          continue;
        }
        if ((int)(e & 255) > b.cnt) { msg = "truncated stream"; return nullptr; }
        b.drop((int)(e & 255));
        if (kind == K_EOB) break;
        if (kind != K_LEN) { msg = "invalid literal/length code"; return nullptr; }
        const int xl = (int)((e >> 12) & 15);
        const uint32_t len = (e >> 16) + b.take(xl);
        uint32_t d = dist[b.peek(DBITS)];
        if (((d >> 8) & 15) == K_SUB) { b.drop(DBITS); d = dist[(d >> 16) + b.peek((int)((d >> 12) & 15))]; }
        if (((d >> 8) & 15) != K_LEN) { msg = "invalid distance code"; return nullptr; }
        b.drop((int)(d & 255));
        const int xd = (int)((d >> 12) & 15);
        if (b.cnt < xd) { b.refill(); if (b.cnt < xd) { msg = "truncated stream"; return nullptr; } }
        const uint32_t distance = (d >> 16) + b.take(xd);
        if ((size_t)(out - base) < distance) { msg = "distance reaches before the stream's start"; return nullptr; }
        if ((size_t)(outEnd - out) < len) { msg = "more text than the output range holds"; fullAt(out); return nullptr; }
        const uint8_t *from = out - distance;
        if (distance >= 8 && (size_t)(outEnd - out) >= (size_t)len + 8) {
          uint8_t *o = out;
          const uint8_t *f = from, *stop = out + len;
          do { uint64_t w; memcpy(&w, f, 8); memcpy(o, &w, 8); o += 8; f += 8; } while (o < stop);
          out += len;
        } else if (distance == 1) {
          memset(out, *from, len);
          out += len;
        } else {
          for (uint32_t i = 0; i < len; ++i) out[i] = from[i];
          out += len;
        }
      }
    }
  blockDone:
    if (b.cnt < 0 || b.under) { msg = "truncated stream"; return nullptr; }
    if (pg) pg->produced.store(progressBase + (uint64_t)(out - dst0), std::memory_order_release);
    if (bfinal) return out;
  }
}

}  // namespace

// every member of the gzip file src[0 .. srcLen) -> dst[0 .. cap); *outLen = bytes of text.  0 = fine; otherwise err says what is wrong.
// progressBase: what the published offsets start from (a file behind others in one text range); finishes: the progress ends with this file.
// The trailer's CRC-32 is NOT checked here (the caller runs it over the text behind the decoder, host/reads.cpp); its length field is.
int gzInflateAll(const uint8_t *src, size_t srcLen, uint8_t *dst, size_t cap, GzProgress *pg, size_t *outLen, uint32_t *lastCrc, size_t *members, std::string &err, uint64_t progressBase,
                 bool finishes) {
  const uint8_t *p = src, *end = src + srcLen;
  uint8_t *out = dst, *outEnd = dst + cap;
  size_t nMembers = 0;
  auto fail = [&](const char *m) { err = std::string("gzip: ") + m; if (pg) pg->state.store(-1, std::memory_order_release); return -1; };
  while (p < end) {
    if (nMembers && *p == 0) { ++p; continue; }  // zero padding behind the last member (tar, some writers)
    // bytes behind a complete member that do not start another one end the data, as they do for zlib's gzread (gz_look with direct == 0:
    // "trailing garbage is ignored"), which is what the reference reads through (ReadFiles.hpp, kseq.h) -- ADVICE round 5
    if (nMembers && (end - p < 2 || p[0] != 0x1f || p[1] != 0x8b)) break;
    if (end - p < 18) return fail(nMembers ? "truncated header" : "file too short");
    if (p[0] != 0x1f || p[1] != 0x8b) return fail("not a gzip file");
    if (p[2] != 8) return fail("unknown compression method");
    const uint8_t flg = p[3];
    p += 10;
    if (flg & 4) { if (end - p < 2) return fail("truncated header"); const size_t xlen = p[0] | (p[1] << 8); p += 2; if ((size_t)(end - p) < xlen) return fail("truncated header"); p += xlen; }
    if (flg & 8) { while (p < end && *p) ++p; if (p >= end) return fail("truncated header"); ++p; }
    if (flg & 16) { while (p < end && *p) ++p; if (p >= end) return fail("truncated header"); ++p; }
    if (flg & 2) { if (end - p < 2) return fail("truncated header"); p += 2; }
    Bits b{p, end};
    const char *msg = nullptr;
    uint8_t *memberStart = out;
    uint8_t *o = inflateRaw(b, memberStart, out, outEnd, pg, dst, progressBase, msg);
    if (!o) return fail(msg);
    // back to whole bytes: the bit buffer holds cnt bits that were loaded but not used
    b.drop(b.cnt & 7);
    p = b.in - (b.cnt >> 3);
    if (end - p < 8) return fail("truncated trailer");
    const uint32_t crc = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24), isize = p[4] | (p[5] << 8) | (p[6] << 16) | ((uint32_t)p[7] << 24);
    if ((uint32_t)(o - memberStart) != isize) return fail("length check of a member failed");
    if (lastCrc) *lastCrc = crc;
    p += 8;
    out = o;
    ++nMembers;
    if (pg) { std::lock_guard<std::mutex> g(pg->m); pg->members.push_back({progressBase + (uint64_t)(out - dst), crc}); }
  }
  if (!nMembers) return fail("empty file");
  if (outLen) *outLen = (size_t)(out - dst);
  if (members) *members = nMembers;
  if (pg) { pg->produced.store(progressBase + (uint64_t)(out - dst), std::memory_order_release); if (finishes) pg->state.store(1, std::memory_order_release); }
  return 0;
}

}  // namespace t1k
