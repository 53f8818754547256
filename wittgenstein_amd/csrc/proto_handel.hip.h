// Handel (P/Handel.java) as a resident device protocol.
//
// State layout (HBM):
//   bit rows  HLevel l's bitsets (totalIncoming, lastAggVerified, verifiedIndSignatures, toVerifyInd, finishedPeers
//             :373-394) only ever hold ids of the level's aligned sibling block of 2^(l-1) ids (allSigsAtLevel :671-684),
//             and the blocks of different levels are disjoint, so W = N/64 words per kind hold all levels. Two arrays,
//             both LEVEL-major (an event works on ONE level), cut by WHO touches a set (round 5):
//               rows  [N][W][3]  TI, LA, VI — the sets updateVerifiedSignatures / checkSigs STREAM (a level's three
//                                side by side, nw words each: h_row);
//               drows [N][W][3]  what a DELIVERY touches: one bit of `from` in each of SEEN, FP (finishedPeers) and BUMP,
//                                the three words holding it side by side in one 24-byte piece (h_dword) — one line per
//                                delivery whatever the level's width, written by no-return atomics, where a level of
//                                >= 512 ids used to cost a line each for VI, TV and FP.
//             toVerifyInd is not stored: it is set at a delivery unless verifiedInd has the sender (:779-781) and cleared
//             by the sender's updateVerifiedSignatures (:701), which sets verifiedInd (:704) for good — so
//             toVerifyInd = SEEN & ~verifiedInd, SEEN = senders of which a SendSigs was accepted; nothing reads the set but
//             the read-back, which computes it. BUMP: senders whose reception rank checkSigs has bumped (see ranks).
//             totalOutgoing of level l is always the union of
//             totalIncoming of levels < l (:728-731) = the node's OWN aligned block (h_word gathers it).
//             totalIncoming is never READ as a row by the checkSigs / updateVerifiedSignatures kernels: it always equals
//             lastAggVerified | verifiedIndSignatures (level 0 starts with the own bit in all three, :413-421; an update
//             sets `from` in VI and TI together, :705-713, or rebuilds TI as LA | VI, :722-724), so those kernels read two
//             rows and OR them — a quarter of their row traffic. The row is still WRITTEN (dissemination snapshots it).
//   ranks     receptionRanks (:285) and the emission lists (:510-522), in one of two forms:
//             MATRIX   ranks [N][N] int32 + peers [N][N-1] ids (16-bit up to 65 536 nodes) — host-built init(), sharded
//                      engines, the attacks' runs, more than 65 536 nodes;
//             CARRIED  (init() on the device, unsharded, no attack, <= 65 536 nodes; round 5) no matrix: a rank is its
//                      INITIAL value — a seeded shuffle, fixed by init() — plus nodeCount per time checkSigs chose that
//                      sender (:825-828). The initial value is what the emission lists were sorted by (:991-1013), so
//                      the SENDER has it: peersR [N][N-1] = id | rank << 16, the rank travels in the message word
//                      (bits 6..21) or, for a fast-path envelope, in the destination word's upper half
//                      (EngineDev::destTagged). The receiver adds its bumps: the BUMP bit of (level, from) in the line
//                      the delivery touches anyway says whether there are any; the count is in a small per-node table
//                      (HandelState::bump). One random line of a 4.3 GB matrix less per delivery, 2.1 GB less per copy.
//   queues    toVerifyAgg (:385): per (node, level) one queue record (h_qrec: length, slots in use, the list
//             in list order — rank, signer, slot per entry — in ONE line for the usual short list) and up to
//             Q signature slots sig[2^(l-1) bits] in a private slab; a slot stays allocated while a registered
//             updateVerifiedSignatures task still references it (:833-836).
//
// Who runs what (a node visit is a chain of dependent memory round trips — the kernels' throughput is
// the number of visits IN FLIGHT over the length of that chain, so the work is cut by how many lanes a visit can use and
// by how many registers a kernel has to hold):
//   k_handel_lane   one LANE per node: the nodes whose events of the ms (<= 4, read from the node's inbox line) are
//                   SendSigs messages (onNewSig :757-790, the hops of fast-path envelopes included; payloads wider than
//                   one word become jobs of k_handel_copy, one wavefront each) and at most one updateVerifiedSignatures
//                   task (:690-754) — of a level whose block is <= 16 words (applied by the lane), or of a wider level as
//                   the node's last event (an item of k_handel_update) — 64 visits in flight per wavefront. It also sorts
//                   every other node into the lists of the kernels below.
//   k_handel_update one WAVEFRONT per wide updateVerifiedSignatures (blocks of 32 .. 256 words), 6 waves/SIMD.
//   k_handel_dissem one WAVEFRONT per node whose FIRST event is its dissemination (:331-343): that event only, 8 waves/SIMD.
//   k_handel_wave   one WAVEFRONT per node, lanes = 64-bit words of the level block: everything else — dissemination
//                   behind other events, nodes with more than four events, the rest of a visit the kernels above handed
//                   on (`skip`), the fast-path sends (:738-749) a lane deferred (remaining_peers is a wave-parallel scan),
//                   and every visit of a run with an attack.
//   checkSigs (:796-837) runs per (node, LEVEL) item — bestToVerify of one level is independent of the other levels:
//   k_handel_cond_pre lists the items, k_handel_a1 curates a level's list and records its candidate (one lane per item
//   for blocks <= 16 words, one wavefront per item beyond: h_best_wave), the scan + k_handel_cond_a2 draw among a
//   node's candidates.
// byzantineSuicide / hiddenByzantine (:538-559, 577-584, 688-694, 813-817, 840-917) are resident in separate instantiations
// (HandelProtoT<true>, k_handel_wave / k_handel_a1 / k_handel_cond_a2 <.., true>, k_handel_hidden): DESIGN.md §3.13.
// init() (:957-1014) runs on the device too (k_handel_init_*, at the end of this file): DESIGN.md §3.11a.
#pragma once
#include "engine_kernels.hip.h"

namespace wg {

constexpr int H_PEND = 4;          // outstanding updateVerifiedSignatures tasks per node
constexpr uint32_t H_TASK_DISSEMINATION = 0;
constexpr uint32_t H_TASK_UPDATE = 1;
// (a queue record — HandelState::qrec — is head, valid mask, Q entries, bad mask in whole 64-byte lines: qStride / qBad)
constexpr int H_QVALID = 2;        // word of the record: bit `slot` = the evaluation cached for that slot's entry (qcache) still holds
constexpr int H_QENT = 3;          // first entry of the list (the head, the valid mask and five entries are ONE 64-byte line)
constexpr int H_LANE_NW = 16;      // level blocks of up to this many 64-bit words are worked on by ONE lane, which streams
                                   // them two words a load (levels <= 11); a wavefront per item spends ~ 25 wave-level memory
                                   // instructions on ONE item, and that instruction rate is what bounds those kernels
constexpr int H_UPD_NW = 256;      // ... and up to this many by k_handel_update's four words a lane (levels <= 15); beyond: k_handel_wave

// The argument word of an updateVerifiedSignatures task (Rec::w3): everything the task needs to issue its loads —
// the pending-table entry it owns, the level, the queue slot and the signer (SigToVerify.from) — so that the visit does
// not wait for the node's header before it can address the level's rows. N <= 2^19.
__device__ __forceinline__ uint32_t h_update_arg(int pk, int lv, int slot, int32_t from) {
  return (uint32_t)pk | ((uint32_t)lv << 2) | ((uint32_t)slot << 7) | ((uint32_t)from << 13);
}
#define H_ARG_PK(a) ((int)((a) & 3u))
#define H_ARG_LV(a) ((int)(((a) >> 2) & 31u))
#define H_ARG_SLOT(a) ((int)(((a) >> 7) & 63u))
#define H_ARG_FROM(a) ((int32_t)((a) >> 13))

struct HandelState {
  wg_handel_params p;
  int32_t N, L, W, Q;
  GP<uint64_t> rows;                      // [N][W][3] TI, LA, VI, level-major: see h_row
  GP<uint64_t> drows;                     // [N][W][3] SEEN, FP, BUMP per 64 ids, level-major: see h_dword
  GP<int32_t> ranks;                      // [N][N]; NULL: the ranks are CARRIED by the senders (file header)
  // CARRIED form: the emission lists with the receiver's initial rank of the sender beside each id, and the receivers' bumps:
  // bump[node][bumpCap] entries count << 16 | from (0: free) — indexed by `from` when bumpCap == N, else open addressing from
  // h_bump_slot(from); checkSigs adds to it (k_handel_cond_a2), a delivery reads it only when the BUMP bit of its sender is set
  GP<const uint32_t> peersR;              // [N][N-1] id | rank << 16
  GP<uint32_t> bump;                      // [N][bumpCap]
  int32_t bumpCap;                        // a power of two <= N (wg_config.rank_bump_cap; a full table is a loud error)
  // byzantineSuicide (P/Handel.java:64-69): HNode.blacklist as one N-bit row per node (:289); HLevel.suicideBizAfter
  // (:406) is the header plane HP_SPARE0, SigToVerify.badSig the record word qBad. atk == 0: none of it is touched
  GP<uint64_t> blacklist;                 // [N][W] (byzantineSuicide only)
  int32_t atk;                            // 1 byzantineSuicide, 2 hiddenByzantine
  int32_t laneNw;                         // H_LANE_NW, or WG_LANE_NW (tests: the wave-per-item paths on networks the emulator can run)
  // emission lists [N][N-1] (:510-522), never written after init(): 16-bit ids when N <= 65536 (half the bytes of the
  // second-largest array of a copy — more resident copies per GPU), 32-bit otherwise; read through h_peer()
  GP<const uint16_t> peers16;
  GP<const int32_t> peers32;
  // toVerifyAgg slots per (node, level): Q for the levels whose block is below 16 words, Qw (wg_config.queue_cap_wide)
  // for the wide ones — their queues stay short (a few entries: one sender per period and level) while a slot is up to
  // 2 KB, so a flat capacity spent gigabytes on slots that are never used
  int32_t Qw;
  // Node header: every scalar of a node and its per-level scalars in ONE record of hdrStride 32-bit words
  // (array of structs). A wave-per-node visit reads it as one 16-byte-per-lane instruction; a lane-per-node visit
  // reads the two or three 16-byte pieces it needs:
  //   [HH_ADDED .. HH_SIGCHK]    addedCycle, sigQueueSize, msgFiltered, startAt, nodePairingTime, currWindowSize, sigsChecked
  //   [HH_TOTAL]                 sum over the levels of |totalIncoming| (what `cur.cardinality()` of :745 is after the loop)
  //   [HH_DONE_LO, HH_DONE_HI]   Node.doneAt, mirrored from NodeArrays::doneAt (written through when it changes)
  //   [HH_PEND +4] [HH_PENDFROM +4]  outstanding updateVerifiedSignatures tasks: valid<<31 | level<<8 | slot ; from
  //   [HH_NRECV .. HH_BSENT]     Node.msgReceived / msgSent (32 bits) and bytesReceived / bytesSent (64 bits) of this
  //                              protocol's deliveries and sends (C/Network.java:476-477,611-612): they change with the words
  //                              above, in the same line — not as four atomics into four more arrays (read back: node_counter)
  //   [HH_QMASK]                 bit l: level l's verification queue is not empty (what k_handel_cond_pre looks at)
  //   [HH_QDIRTY]                bit l: level l's queue holds an entry whose evaluation is not cached (see qcache)
  //   [HH_LV + l*8 + plane]      level-major: the scalars of HLevel l side by side (32 bytes, two levels a 64-byte line) —
  //                              posInLevel, |totalIncoming|, |lastAggVerified|, |verifiedInd|, checkSigs' candidate of this
  //                              edge (signer << 8 | queue slot; valid where candMask[node] has the level's bit: written by
  //                              k_handel_a1, read by k_handel_cond_a2), outgoingFinished; LS = 16 or 32 >= L levels.
  GP<uint32_t> hdr;
  int32_t LS, lsShift, hdrStride;
  // ConditionalTask.minStartTime and the epoch in which the task last left nextMessage()'s copy, two words a node, dense:
  // k_handel_cond_pre looks at every node every ms — 8 bytes of a coalesced stream instead of a line of the record
  GP<uint32_t> ct;
  // toVerifyAgg of (node, level): a queue record of qStride 64-bit words — [0] list length, [1] signature slots in use,
  // [H_QVALID] slots whose cached evaluation holds, [H_QENT + i] entry i in list order: rank << 32 | signer << 8 | slot.
  // Length, slots, valid mask and the first five entries are ONE 64-byte line: what a delivery appends to and checkSigs
  // walks (it used to be a line in each of three arrays)
  GP<uint64_t> qrec;                      // [N][L][qStride]
  // (a listed entry owns one of the level's <= Q signature slots, so a list never holds more than Q entries: the record is
  // head + valid mask + Q entries + bad mask, in whole 64-byte lines — 40 words at the default Q = 32, not H_QREC = 72)
  int32_t qStride, qBad;                  // words of a record; the word of the bad-signature mask (H_QENT + Q)
  // checkSigs made incremental (round 4). bestToVerify (:570-634) asks of every listed signature its sizeIfIncluded
  // (:532-540) and its score (:655-668): functions of the signature — fixed when onNewSig lists it — and of the level's
  // totalIncoming / verifiedIndSignatures / lastAggVerified, which change only when an updateVerifiedSignatures of THAT
  // level runs (:690-754). So an entry's evaluation — keep (sizeIfIncluded > |totalIncoming|) and its score, were it
  // inside the window — is computed once and kept per queue slot: qcache[(node * L + l) * QC + slot] = keep | score << 1,
  // valid while bit `slot` of the record's H_QVALID word is set. onNewSig clears the bit of the slot it takes, an
  // updateVerifiedSignatures that changes one of the level's three sets clears the whole word; the window, the ranks and
  // the choice among the entries are evaluated afresh at every checkSigs, as before. A node's header word HH_QDIRTY has
  // bit l set while level l holds an entry without a valid evaluation: k_handel_cond_pre routes the clean levels' items
  // (every evaluation cached: one record line + one cache line, no row, no signature) to a lane each whatever the width
  // of their block, and clears the word — k_handel_a1 leaves every entry of the edge's items evaluated.
  GP<uint32_t> qcache;                    // [N][L][QC]
  int32_t QC;                             // cache words per (node, level) = queue_cap
  GP<uint64_t> qsig;                      // per level l: [N][Q][nw(l)] at qsigOff[l]
  unsigned long long qsigOff[MAX_LEVELS];
  // dissemination snapshots (SendSigs.sigs = totalOutgoing.clone(), :254): a node disseminates exactly once
  // per aligned window of `period` ms, so its snapshot lives at a computed address — no allocation:
  //   snap[((t / period) % snapNb) * N + node][0 .. snapStride)   the own block of the highest open level;
  //   the lower levels' blocks are sub-ranges of it (see dissemination)
  GP<uint64_t> snap;
  uint32_t snapNb, snapStride;
  // conditional-task phase scratch: the (node, level) items of this edge — node | level << 24 — by the lanes an item uses
  GP<uint32_t> itemsLane, itemsWave;      // [N * L] each
  GP<uint32_t> itemCount;                 // [2] lane items, wave items (reset by k_handel_cond_a2)
  // SendSigs payloads wider than one word that k_handel_lane delivered: copied by k_handel_copy, one wavefront a job
  GP<CopyJob> jobs;                       // [maxEvents]
  // updateVerifiedSignatures tasks of the wide levels that are their node's last event of the ms: {node, event, argument,
  // vflags}, applied by k_handel_update one wavefront each (the node's SendSigs before them by k_handel_lane)
  GP<U4> itemsUpd;                        // [N]
  GP<uint32_t> updCount;                  // [1] (reset with jobCount)
  // nodes whose first event is a wide update with only plain deliveries behind it: k_handel_update applies the update,
  // k_handel_lane2 — one lane per node — the deliveries: {node, vflags << 8 | events, the update's event, 1 if k_handel_update
  // handed the whole visit to k_handel_wave instead}
  GP<U4> itemsTrail;                      // [N]
  GP<uint32_t> trailCount;                // [1] (reset with jobCount)
  // ... and behind a dissemination that was the node's first event (k_handel_dissem applies the task, a second launch of
  // k_handel_lane2 the deliveries — in the ms in which every node disseminates a quarter of the nodes have some)
  GP<U4> itemsTrail2;                     // [N]
  GP<uint32_t> trail2Count;               // [1] (reset with jobCount)
  // nodes whose FIRST event of the ms is their dissemination task: {node, vflags << 8 | events, event, its inbox word 0},
  // applied by k_handel_dissem one wavefront each (the lean kernel of the millisecond in which every node disseminates);
  // the node's later events follow in k_handel_wave (skip = 1)
  GP<U4> itemsDis;                        // [N]
  GP<uint32_t> disCount;                  // [1] (reset with jobCount)
  GP<uint32_t> jobCount;                  // [1] (reset by k_handel_cond_pre of the edge that follows)
  // ... the payloads of at most H_JOB_SMALL words apart: k_handel_copy moves eight of them per wavefront
  GP<CopyJob> jobsSmall;                  // [maxEvents]
  GP<uint32_t> jobSmallCount;             // [1] (reset with jobCount)
  GP<uint32_t> candMask;                  // [N] bit l: level l has a candidate at this edge (0 for a node whose task does not run)
  GP<uint32_t> cleanMask;                 // [N] ... of which: clean levels, answered from their summary by k_handel_cond_pre (no item)
  // sharded engines: how many levels of every node have a candidate at this edge, one BYTE per node (four nodes an int32
  // word) — what the other shards need of candMask to number the draws (exchange 5: N / 4 words instead of N)
  GP<int32_t> xcand;                      // [N / 4 + 1]; NULL when not sharded
  GP<uint32_t> condList;                  // drawing nodes in id order
  GP<int32_t> drawVal;                    // [N]
  // node-range sharding (wg_shard_configure): this engine holds the per-node rows above only for the nodes
  // [lo, hi) (the row pointers are biased so that they are still indexed by node id); 0 / N when not sharded
  int32_t lo, hi;
  GP<const uint64_t> ones;                // [W] all-ones words: the payload of a sharded fast-path send (see snapshot_outgoing)
  GP<uint32_t> snapIdx;                   // [maxEvents] row of a dissemination event in the snapshot exchange image
  GP<uint32_t> nSnap;                     // [1]
  GP<int32_t> xsnap;                      // [xsnapRows][snapStride * 2] exchange image of this ms's dissemination snapshots
  uint32_t xsnapRows;
  // ... or OWNER-DIRECTED (round 5; an engine with an all-to-all transport, Engine::has_alltoall): a dissemination's snapshot
  // is read by the receivers of its messages only, and a level-l receiver sits in the sender's level-l sibling block
  // (:667-680) — for all but the top log2(shards) levels that is the sender's own shard. The sender appends the sub-row a
  // message of ANOTHER shard's node points at to that shard's region of xout, in self-describing chunks — word 0 = ring
  // offset | words << 32, then up to H_XWORDS words — by one atomic per message; the regions travel by all-to-all and the
  // receiver copies every chunk to the same offset of its own ring (k_handel_xunpack). What used to reach every shard (the
  // whole copied row, 8 KB at 131 072 nodes) reaches the one shard that reads it, at the width its level needs.
  int32_t xS, xMe;                        // shards / this shard (xS <= 1: not directed)
  uint32_t xChunkCap;                     // chunks a destination's region holds (overflow: ERR_PAYLOAD, loud)
  GP<uint64_t> xout;                      // [xS][xChunkCap][H_XCHUNK]
  GP<uint32_t> xoutCount;                 // [xS] chunks appended for each destination in this ms
  GP<uint64_t> xin;                       // [xS * xChunkCap][H_XCHUNK] what the other shards sent, back to back
  GP<int32_t> xcounts;                    // [xS][xS] chunks from shard r to shard d (summed across shards before the all-to-all)
};
constexpr int H_XWORDS = 16, H_XCHUNK = 1 + H_XWORDS;
// the shard that owns `node`: shard k holds [N * k / S, N * (k + 1) / S) (Engine::ensure_device)
__device__ __forceinline__ int h_owner(const HandelState& s, int32_t node) {
  int k = (int)(((long long)node * s.xS) / s.N);
  while ((long long)s.N * k / s.xS > node) k--;
  while ((long long)s.N * (k + 1) / s.xS <= node) k++;
  return k;
}

enum HandelHdr : int { HH_ADDED = 0, HH_SIGQ = 1, HH_FILT = 2, HH_START = 3, HH_PAIR = 4, HH_WINDOW = 5, HH_SIGCHK = 6,
                       HH_TOTAL = 7, HH_NRECV = 8, HH_DONE_LO = 9, HH_DONE_HI = 10, HH_QMASK = 11, HH_BRECV = 12, HH_NSENT = 14,
                       HH_QDIRTY = 15, HH_PEND = 16, HH_PENDFROM = 20, HH_HB_LAST = 24, HH_HB_NOBYZ = 25, HH_BSENT = 30, HH_LV = 32 };
// (HH_HB_*: HNode.hiddenByzantine's `last` — 0 = null, else (signer << 8 | slot) + 1 of the entry it planted — and
// noByzantinePeers, P/Handel.java:840-843)
// (words 0..15, one 64-byte line: everything a SendSigs delivery reads and writes of the node; 16..31: what checkSigs and
// updateVerifiedSignatures add to that)
enum HandelKind : int { HK_TI = 0, HK_LA, HK_VI, HK_COUNT };          // HandelState::rows
enum HandelDKind : int { HD_SEEN = 0, HD_FP, HD_BUMP, HD_COUNT = 3 };  // HandelState::drows
enum HandelPlane : int { HP_POS = 0, HP_CTI, HP_CLA, HP_CVI, HP_CAND, HP_OUTFIN, HP_SPARE0, HP_SPARE1, HP_COUNT };
__device__ __forceinline__ uint32_t WG_G* h_hdr(const HandelState& s, int32_t node) { return s.hdr + (size_t)node * s.hdrStride; }
__device__ __forceinline__ uint32_t WG_G* h_lv(const HandelState& s, int32_t node, int plane, int l) {
  return s.hdr + (size_t)node * s.hdrStride + HH_LV + l * HP_COUNT + plane;
}

// geometry of one level inside a row
struct Lv {
  int32_t size;   // ids in the block = expectedSigs()
  int32_t bw;     // first 64-bit word
  int32_t nw;     // number of words
  uint64_t mask;  // bits of the block inside the word when nw == 1 and size < 64, else ~0
};
__device__ __forceinline__ Lv block_view(int32_t firstId, int32_t size) {
  Lv v;
  v.size = size;
  v.bw = firstId >> 6;
  if (size >= 64) {
    v.nw = size >> 6;
    v.mask = ~0ULL;
  } else {
    v.nw = 1;
    v.mask = ((1ULL << size) - 1ULL) << (firstId & 63);
  }
  return v;
}
// the sibling block: ids whose signatures level l waits for (waitedSigs :424-432)
__device__ __forceinline__ Lv sib_view(int32_t node, int l) {
  if (l == 0) return block_view(node, 1);
  return block_view(((node >> (l - 1)) ^ 1) << (l - 1), 1 << (l - 1));
}
// the node's own block: ids that totalOutgoing of level l can hold
__device__ __forceinline__ Lv own_view(int32_t node, int l) { return block_view((node >> (l - 1)) << (l - 1), 1 << (l - 1)); }

__device__ __forceinline__ int32_t h_peer(const HandelState& s, size_t idx) {
  if (s.peersR) return (int32_t)(s.peersR[idx] & 0xFFFFu);
  return s.peers16 ? (int32_t)s.peers16[idx] : s.peers32[idx];
}
// ... with the peer's initial reception rank of this sender in the upper half (CARRIED form; 0 there in the MATRIX form):
// what a SendSigs to that peer carries — as it is in a destination word (EngineDev::destTagged), h_msg_rank in a message word
__device__ __forceinline__ uint32_t h_peer_tagged(const HandelState& s, size_t idx) {
  if (s.peersR) return s.peersR[idx];
  return (uint32_t)(s.peers16 ? (int32_t)s.peers16[idx] : s.peers32[idx]);
}
constexpr int H_MSG_RANK_SHIFT = 6;  // SendSigs message word: level (5 bits) | levelFinished << 5 | initial rank << 6 (CARRIED form)
__device__ __forceinline__ uint32_t h_msg_rank(const HandelState& s, uint32_t tagged) { return s.peersR ? (tagged >> 16) << H_MSG_RANK_SHIFT : 0u; }
__device__ __forceinline__ int32_t h_tag_id(const HandelState& s, uint32_t tagged) { return (int32_t)(s.peersR ? tagged & 0xFFFFu : tagged); }
__device__ __forceinline__ int h_nw(int l) { return l == 0 ? 1 : ((1 << (l - 1)) >= 64 ? (1 << (l - 1)) >> 6 : 1); }
__device__ __forceinline__ int h_qcap(const HandelState& s, int l) { return h_nw(l) >= 16 ? s.Qw : s.Q; }
// kind k's words of level l's sibling block: h_nw(l) contiguous words. Per node the W words of a kind are split by level
// — group 0 is the node's own 64-id word (it holds the blocks of the levels 0..6, told apart by Lv::mask), level l >= 7 is
// its sibling block of 2^(l-7) words, and 2^(l-7) words precede it — and inside a group the five kinds follow each other.
__device__ __forceinline__ uint64_t WG_G* h_row(const HandelState& s, int32_t node, int k, int l) {
  const int nw = h_nw(l), before = l <= 6 ? 0 : nw;
  return s.rows + ((size_t)node * s.W + before) * HK_COUNT + (size_t)k * nw;
}
// the delivery-side piece {SEEN, FP, BUMP} of word w of level l's sibling block (24 bytes: one line, for one piece in four
// two); kind k of it is [k]. The same level-major grouping as the rows.
__device__ __forceinline__ uint64_t WG_G* h_dword(const HandelState& s, int32_t node, int l, int w) {
  const int before = l <= 6 ? 0 : h_nw(l);
  return s.drows + ((size_t)node * s.W + before + w) * HD_COUNT;
}
// ... of word x of the N-bit row (any level), as h_word below
__device__ __forceinline__ uint64_t WG_G* h_dword_x(const HandelState& s, int32_t node, int x) {
  const uint32_t diff = (uint32_t)x ^ (uint32_t)(node >> 6);
  if (diff == 0) return s.drows + (size_t)node * s.W * HD_COUNT;
  const int nw = 1 << (31 - __clz(diff));
  return s.drows + ((size_t)node * s.W + nw + (x & (nw - 1))) * HD_COUNT;
}
// finishedPeers.get(p) of level l whose sibling block starts at word fbw
__device__ __forceinline__ bool h_finished(const HandelState& s, int32_t node, int l, int fbw, int32_t p) {
  const uint64_t w = __hip_atomic_load(h_dword(s, node, l, (p >> 6) - fbw) + HD_FP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (w >> (p & 63)) & 1ULL;
}
// ---- the receiver's side of a CARRIED rank (h_bump itself: below h_rank_at_delivery): initial value + nodeCount per bump, saturating as :825-828 does
__device__ __forceinline__ uint32_t h_bump_slot(const HandelState& s, int32_t from) {
  return s.bumpCap == s.N ? (uint32_t)from : (((uint32_t)from * 0x9E3779B1u) >> 12) & (uint32_t)(s.bumpCap - 1);
}
__device__ __forceinline__ uint32_t h_bump_count(const HandelState& s, int32_t node, int32_t from) {
  const uint32_t WG_G* tab = s.bump + (size_t)node * s.bumpCap;
  uint32_t h = h_bump_slot(s, from);
  if (s.bumpCap == s.N) return tab[h] >> 16;
  for (int probes = 0; probes < s.bumpCap; probes++) {
    const uint32_t e = tab[h];
    if (e == 0) return 0;
    if ((int32_t)(e & 0xFFFFu) == from) return e >> 16;
    h = (h + 1) & (uint32_t)(s.bumpCap - 1);
  }
  return 0;
}
__device__ __forceinline__ int32_t h_rank_of(const HandelState& s, uint32_t rank0, uint32_t count) {
  const unsigned long long r = (unsigned long long)rank0 + (unsigned long long)count * (unsigned long long)(uint32_t)s.N;
  return r > 0x7FFFFFFFull ? INT32_MAX : (int32_t)r;
}
// receptionRanks[from] as a delivery reads it (:784): the matrix entry, or the carried initial rank + this node's bumps of
// `from` (bumpWord: the BUMP word of the level's piece holding `from`, loaded by the caller with the delivery's other loads)
__device__ __forceinline__ int32_t h_rank_at_delivery(const HandelState& s, int32_t node, int32_t from, uint32_t msg, uint64_t bumpWord) {
  const uint32_t rank0 = (msg >> H_MSG_RANK_SHIFT) & 0xFFFFu;
  if (!((bumpWord >> (from & 63)) & 1ULL)) return (int32_t)rank0;
  return h_rank_of(s, rank0, h_bump_count(s, node, from));
}
// receptionRanks[from] += nodeCount (:825-828) in the CARRIED form: one more bump of (node, from) — by the ONE thread that
// runs the node's checkSigs at this edge (k_handel_cond_a2) — and the BUMP bit of `from` in level l's delivery piece.
// A count beyond 16 bits (65 535 bumps of one sender by one node) or a full table stops the run loudly.
__device__ __forceinline__ void h_bump(const EngineDev& d, const HandelState& s, int32_t node, int l, int32_t from) {
  uint32_t WG_G* tab = s.bump + (size_t)node * s.bumpCap;
  uint32_t h = h_bump_slot(s, from);
  bool done = false;
  for (int probes = 0; probes < s.bumpCap && !done; probes++) {
    const uint32_t e = tab[h];
    if (e == 0 || (int32_t)(e & 0xFFFFu) == from || s.bumpCap == s.N) {
      const uint32_t c = (e >> 16) + 1u;
      if (c > 0xFFFFu) {  // (65 535 bumps of one sender by one node: the count's 16 bits — the rank table's own error, not a generic one)
        set_err(d.g, ERR_RANK_BUMPS);
        return;
      }
      tab[h] = (c << 16) | (uint32_t)from;
      // (saturation :826-828 depends on the initial rank too — any initial rank saturates once count * N alone does)
      if ((unsigned long long)c * (unsigned long long)(uint32_t)s.N > 0x7FFFFFFFull) atomicOr(&d.g->notes, NOTE_RANKS_SATURATED);
      done = true;
    }
    h = (h + 1) & (uint32_t)(s.bumpCap - 1);
  }
  if (!done) {
    set_err(d.g, ERR_RANK_BUMPS);
    return;
  }
  const int w = (from >> 6) - sib_view(node, l).bw;
  atomicOr((unsigned long long*)F(h_dword(s, node, l, w) + HD_BUMP), 1ULL << (from & 63));
}
// word x (index in the N-bit row of kind k, any level) of `node`: which level's block it is follows from where x
// differs from the node's own word index
__device__ __forceinline__ uint64_t WG_G* h_word(const HandelState& s, int32_t node, int k, int x) {
  const uint32_t diff = (uint32_t)x ^ (uint32_t)(node >> 6);
  if (diff == 0) return s.rows + (size_t)node * s.W * HK_COUNT + k;
  const int nw = 1 << (31 - __clz(diff));  // the sibling block of level 7 + log2(nw)
  return s.rows + ((size_t)node * s.W + nw) * HK_COUNT + (size_t)k * nw + (x & (nw - 1));
}
__device__ __forceinline__ int h_msg_size(int l) { return 1 + ((l == 0 ? 1 : (1 << (l - 1))) / 8) + 96 * 2; }  // :256-260

// word j of a level's block is owned by lane j & 63
#define H_FOR_WORDS(v, j) for (int j = (int)WG_LANE; j < (v).nw; j += 64)

__device__ __forceinline__ uint64_t ld_coherent(const uint64_t WG_G* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool row_get(const uint64_t WG_G* row, int32_t id) { return (ld_coherent(row + (id >> 6)) >> (id & 63)) & 1ULL; }
__device__ __forceinline__ void row_set(uint64_t WG_G* row, int32_t id, bool v) {
  int w = id >> 6;
  if ((int)WG_LANE == (w & 63)) {
    uint64_t x = row[w];
    row[w] = v ? (x | (1ULL << (id & 63))) : (x & ~(1ULL << (id & 63)));
  }
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) { return wave_reduce_add64(v); }

// One lane streams a level's block: word j of up to four of the level's arrays at a time, two words a load where the
// block has them (blocks of >= 2 words start 16-byte aligned). F(j, a, b, c, d) is called once per word.
struct alignas(16) V2 {
  uint64_t x, y;
};
template <class F>
__device__ __forceinline__ void h_stream4(const uint64_t WG_G* pa, const uint64_t WG_G* pb, const uint64_t WG_G* pc,
                                          const uint64_t WG_G* pd, int nw, F f) {
  if (nw == 1) {
    f(0, pa[0], pb[0], pc[0], pd[0]);
    return;
  }
#pragma unroll 2
  for (int j = 0; j < nw; j += 2) {
    const V2 a = gld((const V2 WG_G*)(pa + j)), b = gld((const V2 WG_G*)(pb + j));
    const V2 c = gld((const V2 WG_G*)(pc + j)), dd = gld((const V2 WG_G*)(pd + j));
    f(j, a.x, b.x, c.x, dd.x);
    f(j + 1, a.y, b.y, c.y, dd.y);
  }
}

// per-wave LDS mirror of the (node, level) scalars
struct LevelScalars {  // LDS image of a node header: the planes HP_POS..HP_SPARE1 (32 words each), then the scalars
  int32_t pos[32];
  int32_t cTI[32];
  int32_t cLA[32];
  int32_t cVI[32];
  uint32_t cand[32];
  int32_t outFin[32];
  uint32_t spare0[32];
  uint32_t spare1[32];
  uint32_t sc[HH_LV];
  U4 orig[(HH_LV + 8 * 32) / 4];  // the record as it was loaded, 16-byte pieces: store_levels writes back what differs
};
constexpr int H_JOB_SMALL = 64;  // words: a payload of a level of up to 4096 ids (16 / 32 / 64 / 128 measured: profiles/r16i)
// a wide payload delivered by a lane becomes a job of k_handel_copy, listed by size class — one atomic per wavefront and class
// (every lane of the wavefront calls it; job.nw == 0: none)
__device__ __forceinline__ void h_emit_job(const HandelState& s, const CopyJob& job) {
  const int lane = WG_LANE;
  const bool small = job.nw > 0 && job.nw <= H_JOB_SMALL, large = job.nw > H_JOB_SMALL;
  const uint64_t ms = __ballot(small), ml = __ballot(large);
  if (ms) {
    uint32_t jb = 0;
    const int leader = __ffsll((unsigned long long)ms) - 1;
    if (lane == leader) jb = atomicAdd(F(s.jobSmallCount + 0), (uint32_t)__popcll(ms));
    jb = lane_bcast(jb, leader);
    if (small) gst(s.jobsSmall + (jb + __popcll(ms & lanes_lt())), job);
  }
  if (ml) {
    uint32_t jb = 0;
    const int leader = __ffsll((unsigned long long)ml) - 1;
    if (lane == leader) jb = atomicAdd(F(s.jobCount + 0), (uint32_t)__popcll(ml));
    jb = lane_bcast(jb, leader);
    if (large) gst(s.jobs + (jb + __popcll(ml & lanes_lt())), job);
  }
}
constexpr uint32_t H_REF_RING = 0x80000000u;  // payload ref flag: engine payload ring (fast-path sends)
constexpr uint32_t H_REF_ONES = 0xFFFFFFFFu;  // payload ref: the all-ones block (fast-path sends of a sharded engine)
__device__ __forceinline__ const uint64_t WG_G* h_payload(const EngineDev& d, const HandelState& s, uint32_t payload) {
  if (payload & H_REF_RING) return payload == H_REF_ONES ? s.ones : d.payload + (payload & ~H_REF_RING);
  return s.snap + payload;
}
__device__ __forceinline__ uint64_t WG_G* h_sig_ptr(const HandelState& s, int32_t node, int l, int slot) {
  return s.qsig + s.qsigOff[l] + ((size_t)node * h_qcap(s, l) + slot) * (size_t)h_nw(l);
}
__device__ __forceinline__ uint32_t h_pend_word(int l, int slot) { return 0x80000000u | ((uint32_t)l << 8) | (uint32_t)slot; }
__device__ __forceinline__ uint64_t WG_G* h_qrec(const HandelState& s, int32_t node, int l) {
  return s.qrec + ((size_t)node * s.L + l) * (size_t)s.qStride;
}
__device__ __forceinline__ uint64_t h_entry(int32_t rank, int32_t from, int slot) {
  return ((uint64_t)(uint32_t)rank << 32) | ((uint64_t)(uint32_t)from << 8) | (uint32_t)slot;
}
// a record's head: {length, slots in use} as one 16-byte access
struct alignas(16) HQHead {
  uint64_t len, used;
};

// ATK: the byzantineSuicide paths compiled in (a run without the attack uses the instantiation without them)
template <bool ATK>
struct HandelProtoT {
  typedef HandelState State;
  typedef LevelScalars WaveShared;
  // blacklist.get(id) of `node` (:289)
  __device__ static bool blk(const State& s, int32_t node, int32_t id) {
    if (!ATK || s.atk != 1) return false;
    return (ld_coherent(s.blacklist + (size_t)node * s.W + (id >> 6)) >> (id & 63)) & 1ULL;
  }

  // node-scoped registers (wave-uniform) live in this struct for the duration of a node's events
  struct NodeRegs {
    long long doneAt, doneAt0;
    int32_t addedCycle, sigQueueSize, msgFiltered, startAt;
    LevelScalars* ls;  // (the outstanding updateVerifiedSignatures tasks stay in the LDS image: sc[HH_PEND ..], sc[HH_PENDFROM ..])
  };

  __device__ static int msg_size(const State&, uint32_t msg) { return h_msg_size((int)(msg & 31u)); }
  __device__ static int msg_level(uint32_t msg) { return (int)(msg & 31u); }

  // the header fetched ahead of the visit (the pipelined loop of k_handel_wave): 16 bytes a lane, two rounds when L > 16
  struct Pre {
    U4 q0;
  };
  __device__ static Pre prefetch(const State& s, int32_t node) {
    const U4 WG_G* g = (const U4 WG_G*)h_hdr(s, node);
    const int n4 = s.hdrStride >> 2;
    Pre p;
    p.q0 = g[(int)WG_LANE < n4 ? (int)WG_LANE : 0];
    return p;
  }
  // piece i of the record (16 bytes) <-> the LDS image: scalars as they are; a level's two pieces are the planes
  // 0..3 / 4..7 of that level (the image keeps one array per plane: ls->cTI[l], ...)
  __device__ static void scatter_levels(const State&, LevelScalars* ls, int i, const U4 q) {
    const int w = i << 2;
    ls->orig[i] = q;
    if (w < HH_LV) {
      uint32_t* dst = ls->sc + w;
      dst[0] = q.x;
      dst[1] = q.y;
      dst[2] = q.z;
      dst[3] = q.w;
    } else {
      const int r = w - HH_LV;
      uint32_t* dst = (uint32_t*)ls + (((r >> 2) & 1) << 7) + (r >> 3);  // plane (0 or 4) * 32 + level
      dst[0] = q.x;
      dst[32] = q.y;
      dst[64] = q.z;
      dst[96] = q.w;
    }
  }
  __device__ static U4 gather_levels(const LevelScalars* ls, int i) {
    const int w = i << 2;
    U4 q;
    if (w < HH_LV) {
      const uint32_t* src = ls->sc + w;
      q.x = src[0];
      q.y = src[1];
      q.z = src[2];
      q.w = src[3];
    } else {
      const int r = w - HH_LV;
      const uint32_t* src = (const uint32_t*)ls + (((r >> 2) & 1) << 7) + (r >> 3);
      q.x = src[0];
      q.y = src[32];
      q.z = src[64];
      q.w = src[96];
    }
    return q;
  }
  __device__ static void node_begin_pre(Ctx& c, const State& s, NodeRegs& r, LevelScalars* ls, const Pre& p) {
    const int n4 = s.hdrStride >> 2;
    __builtin_amdgcn_wave_barrier();  // the previous visit's store_levels has read the image
    if ((int)WG_LANE < n4) scatter_levels(s, ls, (int)WG_LANE, p.q0);
    if ((int)WG_LANE + 64 < n4)  // (L > 16: the record's tail is fetched here, not ahead)
      scatter_levels(s, ls, (int)WG_LANE + 64, ((const U4 WG_G*)h_hdr(s, c.node))[(int)WG_LANE + 64]);
    __builtin_amdgcn_wave_barrier();
    regs_from_image(r, ls);
  }
  __device__ static void regs_from_image(NodeRegs& r, LevelScalars* ls) {
    // (every lane reads the same LDS words: readfirstlane tells the compiler the values are wave-uniform — SGPRs)
    const uint32_t* h = ls->sc;
    r.doneAt = r.doneAt0 = (long long)((unsigned long long)WG_READFIRST(h[HH_DONE_LO]) | ((unsigned long long)WG_READFIRST(h[HH_DONE_HI]) << 32));
    r.addedCycle = (int32_t)WG_READFIRST(h[HH_ADDED]);
    r.sigQueueSize = (int32_t)WG_READFIRST(h[HH_SIGQ]);
    r.msgFiltered = (int32_t)WG_READFIRST(h[HH_FILT]);
    r.startAt = (int32_t)WG_READFIRST(h[HH_START]);
    r.ls = ls;
  }
  __device__ static void node_end(Ctx& c, const State& s, NodeRegs& r) {
    const int32_t node = c.node;
    __builtin_amdgcn_wave_barrier();
    if (WG_LANE == 0) {
      uint32_t* h = r.ls->sc;
      h[HH_DONE_LO] = (uint32_t)(unsigned long long)r.doneAt;
      h[HH_DONE_HI] = (uint32_t)((unsigned long long)r.doneAt >> 32);
      h[HH_ADDED] = (uint32_t)r.addedCycle;
      h[HH_SIGQ] = (uint32_t)r.sigQueueSize;
      h[HH_FILT] = (uint32_t)r.msgFiltered;
      if (r.doneAt != r.doneAt0) c.d.nodes.doneAt[node] = r.doneAt;  // Node.doneAt proper (read-back, contIf)
    }
    store_levels(s, node, r.ls);
  }
  // Node.msgReceived / bytesReceived / msgSent / bytesSent of a visit (C/Network.java:476-477,611-612) go into the
  // header's image and are stored with it (node_end), not as atomics into the engine's four counter arrays
  __device__ static void node_counters(Ctx& c, const State&, NodeRegs& r, long long nRecv, long long bRecv) {
    if (WG_LANE == 0) {
      uint32_t* h = r.ls->sc;
      h[HH_NRECV] += (uint32_t)nRecv;
      h[HH_NSENT] += (uint32_t)c.msgSent;
      const unsigned long long br = ((unsigned long long)h[HH_BRECV] | ((unsigned long long)h[HH_BRECV + 1] << 32)) + (unsigned long long)bRecv;
      const unsigned long long bs = ((unsigned long long)h[HH_BSENT] | ((unsigned long long)h[HH_BSENT + 1] << 32)) + (unsigned long long)c.bytesSent;
      h[HH_BRECV] = (uint32_t)br;
      h[HH_BRECV + 1] = (uint32_t)(br >> 32);
      h[HH_BSENT] = (uint32_t)bs;
      h[HH_BSENT + 1] = (uint32_t)(bs >> 32);
    }
  }
  __device__ static void on_message(Ctx& c, const State& s, NodeRegs& r, int32_t from, uint32_t msg, uint32_t payload) {
    KPROF_DECL;
    on_new_sig(c, s, r, from, msg, payload);
    KPROF_COUNT(c.d.g, 4);
    KPROF_MARK(c.d.g, 5);
  }
  __device__ static void on_task(Ctx& c, const State& s, NodeRegs& r, uint32_t word, uint32_t arg) {
    KPROF_DECL;
    if (word == H_TASK_DISSEMINATION) {
      dissemination(c, s, r);
      KPROF_COUNT(c.d.g, 6);
      KPROF_MARK(c.d.g, 7);
    } else {
      update_verified(c, s, r, arg);
      KPROF_COUNT(c.d.g, 11);
      KPROF_MARK(c.d.g, 12);
    }
  }

  // the header back to memory: only the 16-byte pieces that differ from what was loaded (an event changes the scalars'
  // line and its level's: the other lines of the record stay clean in L2 and are never written back to HBM)
  __device__ static void store_levels(const State& s, int32_t node, const LevelScalars* ls) {
    __builtin_amdgcn_wave_barrier();
    U4 WG_G* g = (U4 WG_G*)h_hdr(s, node);
    for (int i = WG_LANE; i < (s.hdrStride >> 2); i += 64) {
      const U4 q = gather_levels(ls, i), o = ls->orig[i];
      if (q.x != o.x || q.y != o.y || q.z != o.z || q.w != o.w) g[i] = q;
    }
  }

  // ---- queue helpers ---------------------------------------------------------------------------
  __device__ static uint64_t WG_G* sig_ptr(const State& s, int32_t node, int l, int slot) { return h_sig_ptr(s, node, l, slot); }
  __device__ static bool slot_pending(const NodeRegs& r, int l, int slot) {
    bool p = false;
#pragma unroll
    for (int k = 0; k < H_PEND; k++) p |= (r.ls->sc[HH_PEND + k] == h_pend_word(l, slot));
    return p;
  }
  // ---- getRemainingPeers (:486-508), wave-parallel but sequentially equivalent ------------------
  // Scans the emission list from posInLevel, 64 peers per step. Accepted peers (not finished) are
  // appended to the dest ring at destOff (want > 1) or returned (want == 1, wave-uniform in *single) — as the list holds
  // them (h_peer_tagged: the peer's rank of this node rides in the upper half in the CARRIED form). Returns the count.
  __device__ static int remaining_peers(Ctx& c, const State& s, LevelScalars* ls, int l, int want, uint32_t destOff,
                                        uint32_t* single) {
    const int32_t node = c.node;
    const int size = 1 << (l - 1);
    const size_t peers0 = (size_t)node * (s.N - 1) + (size - 1);
    const int fbw = sib_view(node, l).bw;  // the level's peers are the ids of its sibling block
    int pos = ls->pos[l];
    const int start = pos;
    int got = 0;
    bool fin = false;
    while (want > 0 && !fin) {
      int len = min(64, size - pos);
      int k = WG_LANE;
      bool in = k < len;
      const uint32_t pt = in ? h_peer_tagged(s, peers0 + pos + k) : 0u;
      const int32_t p = h_tag_id(s, pt);
      bool ok = in && !h_finished(s, node, l, fbw, p);
      if (ATK && ok) ok = !blk(s, node, p);  // ... && !blacklist.get(p.nodeId)  :493
      uint64_t okm = __ballot(ok);
      // a rejected peer whose successor position is `start` finishes the level (:499-503)
      int nextPos = pos + k + 1;
      if (nextPos >= size) nextPos = 0;
      uint64_t finm = __ballot(in && !ok && nextPos == start);
      // stop index: the want-th accepted lane, or the finishing lane, whichever comes first
      int stopAcc = 64, stopFin = finm ? __ffsll((unsigned long long)finm) - 1 : 64;
      if (__popcll(okm) >= want) {
        uint64_t m = okm;
        for (int i = 1; i < want; i++) m &= m - 1;
        stopAcc = __ffsll((unsigned long long)m) - 1;
      }
      int stop = min(stopAcc, stopFin);
      int consumed = stop < 64 ? stop + 1 : len;
      uint64_t take = okm & (consumed >= 64 ? ~0ULL : ((1ULL << consumed) - 1ULL));
      int ntake = __popcll(take);
      const int idx = got + __popcll(take & lanes_lt());
      const bool mine = ok && k < consumed;
      if (single) {  // (want == 1: the one accepted peer, from the lane that holds it)
        const uint64_t sm = __ballot(mine && idx == 0);
        if (sm) *single = lane_bcast(pt, __ffsll((unsigned long long)sm) - 1);
      }
      if (mine && destOff != 0xFFFFFFFFu) c.dest_put(destOff, idx, (int32_t)pt);
      got += ntake;
      want -= ntake;
      pos += consumed;
      if (pos >= size) pos = 0;
      if (stopFin < 64 && stopFin <= stopAcc && stopFin < consumed) fin = true;
    }
    if (WG_LANE == 0) {
      ls->pos[l] = pos;
      if (fin) ls->outFin[l] = 1;
    }
    __builtin_amdgcn_wave_barrier();
    return got;
  }

  // snapshot of totalOutgoing of level l (the node's own block of the TI row) for a fast-path send
  // (irregular: engine payload ring); the periodic dissemination has its own computed slot.
  __device__ static uint32_t snapshot_outgoing(Ctx&, const State&, int) {
    // A fast-path send happens when totalOutgoing of level l is COMPLETE (cur == 2^(l-1), :738-741): its snapshot is the
    // node's whole own block, i.e. all ones under the receiver's level mask — the constant block, as for a complete level's
    // dissemination. (Until round 5 an unsharded engine copied the block into the engine's payload ring: 277 MB of ring per
    // copy and its image, for words that are all ones.)
    return H_REF_ONES;
  }

  // ---- Message.action: SendSigs -> onNewSig (:757-790), one wavefront per node ------------------------------
  __device__ static void on_new_sig(Ctx& c, const State& s, NodeRegs& r, int32_t from, uint32_t msg, uint32_t payload) {
    const int l = (int)(msg & 31u);
    const bool levelFinished = (msg >> 5) & 1u;
    const int32_t node = c.node;
    if (r.doneAt > 0) {
      r.msgFiltered++;
      return;
    }
    if (c.t < r.startAt) return;
    if (ATK && blk(s, node, from)) return;  // :766
    LevelScalars* ls = r.ls;
    // Everything this event reads from HBM depends only on (node, from, payload): issue it all before the
    // first use so the event costs ONE memory round trip (the path is latency-bound).
    const Lv v = sib_view(node, l);
    const int w = (from >> 6) - v.bw;  // the word of `from` inside the level's block
    const uint64_t bit = 1ULL << (from & 63);
    uint64_t WG_G* dp = h_dword(s, node, l, w);  // {SEEN, FP, BUMP} words holding `from`: one 32-byte piece
    const uint64_t seenW = ld_coherent(dp + HD_SEEN), fpW = ld_coherent(dp + HD_FP);
    // receptionRanks[from], read at receive time (:784)
    int32_t rank;
    if (s.ranks)
      rank = s.ranks[(size_t)node * s.N + from];
    else
      rank = h_rank_at_delivery(s, node, from, msg, ld_coherent(dp + HD_BUMP));
    uint64_t WG_G* qr = h_qrec(s, node, l);
    const HQHead qh = gld((const HQHead WG_G*)qr);
    const uint64_t qvalid = qr[H_QVALID];
    const uint64_t WG_G* src = h_payload(c.d, s, payload);
    const int j0 = (int)WG_LANE;
    const bool has0 = j0 < v.nw;
    uint64_t pw0 = 0;
    if (has0) pw0 = src[j0] & v.mask;
    if (WG_LANE == 0) {  // (plain stores: this wavefront is the node's only writer in its launch; an L2 atomic costs several stores)
      if (levelFinished && !(fpW & bit)) dp[HD_FP] = fpW | bit;  // finishedPeers.set(from)
      if (!(seenW & bit)) dp[HD_SEEN] = seenW | bit;              // toVerifyInd.set(from) unless verified: SEEN & ~VI (file header)
    }
    r.sigQueueSize++;
    // toVerifyAgg.add(new SigToVerify(from, level, receptionRanks[from], cs, badSig))
    const unsigned long long used = qh.used;
    const int qc = h_qcap(s, l);
    unsigned long long capMask = qc >= 64 ? ~0ULL : ((1ULL << qc) - 1ULL);
    unsigned long long freeM = ~used & capMask;
    const int len = (int)qh.len;
    if (freeM == 0 || len >= 64) {
      if (WG_LANE == 0) set_err(c.d.g, ERR_QUEUE_CAP);
      return;
    }
    int slot = __ffsll(freeM) - 1;
    uint64_t WG_G* dst = sig_ptr(s, node, l, slot);
    if (has0) dst[j0] = pw0;
    for (int j = j0 + 64; j < v.nw; j += 64) dst[j] = src[j];
    __builtin_amdgcn_wave_barrier();  // every lane has read the record's head before lane 0 replaces it
    if (WG_LANE == 0) {
      qr[H_QENT + len] = h_entry(rank, from, slot);
      HQHead nh;
      nh.len = (uint64_t)(len + 1);
      nh.used = used | (1ULL << slot);
      gst((HQHead WG_G*)qr, nh);
      if ((qvalid >> slot) & 1ULL) qr[H_QVALID] = qvalid & ~(1ULL << slot);  // the slot's cached evaluation was its previous entry's
      if (ATK) qr[s.qBad] &= ~(1ULL << slot);  // badSig = false (ssigs.badSig is never set by a sender :786)
      ls->sc[HH_QMASK] |= 1u << l;
      ls->sc[HH_QDIRTY] |= 1u << l;
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- PeriodicTask: dissemination (:331-343) -> HLevel.doCycle (:474-484) ------------------------
  // The reference walks the levels one after the other; done literally that is one chain of dependent
  // HBM round trips per level (peer id -> its finishedPeers bit -> the snapshot's words). Here lane l
  // resolves level l's isOpen() and first candidate peer, so all levels' lookups are two round trips in
  // total, and the snapshots of all open levels are one flat copy. A level whose first candidate is a
  // finished peer (the scan of getRemainingPeers :486-508 has to go on) takes the sequential path.
  __device__ static void dissemination(Ctx& c, const State& s, NodeRegs& r) {
    if (r.doneAt > 0) {
      if (r.addedCycle > 0)
        r.addedCycle--;
      else
        return;
    }
    LevelScalars* ls = r.ls;
    const int lane = WG_LANE;
    const int32_t node = c.node;
    // |totalOutgoing| of level l = sum of |totalIncoming| below l
    const int cti = lane < s.L ? ls->cTI[lane] : 0;
    const int incl = (int)wave_incl_scan32((uint32_t)cti);
    const int below = incl - cti;
    bool open = false, fin = false;
    int32_t cand = 0, cand2 = 0;
    uint32_t candT = 0, cand2T = 0;  // ... as the list holds them (h_peer_tagged): what the send carries
    int myPos = 0;
    bool two = false, fin2 = true;
    const int mySize = (lane >= 1 && lane < s.L) ? 1 << (lane - 1) : 0;
    if (mySize) {
      open = !ls->outFin[lane] && (c.t >= (lane - 1) * s.p.levelWaitTime || below == mySize);  // isOpen :458-472
      if (open) {
        myPos = ls->pos[lane];
        const size_t at = (size_t)node * (s.N - 1) + (mySize - 1) + myPos;
        candT = h_peer_tagged(s, at);
        cand = h_tag_id(s, candT);
        // ... and the peer after it, in the same round trip: when the first candidate is a finished peer the scan of
        // getRemainingPeers usually ends at the next one. Only where neither a wrap of posInLevel nor the end-of-scan test
        // (`posInLevel == start`, :499-503) can fall between the two: a level of more than two peers, not at its last position
        two = mySize > 2 && myPos + 1 < mySize;
        if (two) {
          cand2T = h_peer_tagged(s, at + 1);
          cand2 = h_tag_id(s, cand2T);
        }
      }
    }
    KPROF_DECL;
    const uint64_t openM = __ballot(open);
    if (!openM) return;
    // snapshots (SendSigs.sigs = totalOutgoing.clone() :254): totalOutgoing of level l is the node's own aligned
    // block of 2^(l-1) ids, and those blocks are nested — so ONE copy of the highest open level's block serves every
    // level: level l's message points at its sub-range (the receiver masks single-word blocks with its level mask).
    // The block's words are requested HERE, before the candidates' finished bits are waited for: the two round trips
    // overlap (in the millisecond in which every node disseminates this copy is most of the pass's traffic).
    const uint32_t win = ((uint32_t)c.t / (uint32_t)s.p.disseminationPeriodMs) % s.snapNb;
    const uint32_t refBase = (win * (uint32_t)s.N + (uint32_t)node) * s.snapStride;
    // A level whose totalOutgoing is COMPLETE (every id of its block: `below == mySize`) sends the constant all-ones block
    // instead of a copy (the receiver masks it with its level's mask, as it does a snapshot): complete levels are the low
    // ones (an incomplete block makes every block above it incomplete), so the copy is of the highest open INCOMPLETE
    // level's block — none at all for a node that holds its whole half, which is every node for the last third of a run.
    // Those messages' payload reads then hit 2 KB that never leave L2. A sharded engine ships to the other shards only what
    // was copied: the event's result carries the row's width (EV_SNAP_*), nothing for a node without an incomplete open level.
    const bool complete = mySize != 0 && below == mySize;
    const uint64_t copyM = __ballot(open && !complete);
    Lv tv = own_view(node, copyM ? 63 - __clzll((unsigned long long)copyM) : 1);
    if (!copyM) tv.nw = 0;
    c.evFlags = tv.nw ? ev_snap_code((uint32_t)tv.nw) << EV_SNAP_SHIFT : 0u;
    // (two words a lane: the kernel is bound by its wave-level memory instructions. A pair of consecutive block words
    // sits side by side in one level's group — except the pair made of the node's own word and its level-7 sibling)
    V2 sv[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int j = 2 * (q * 64 + lane);
      sv[q].x = sv[q].y = 0;
      if (tv.nw == 1) {
        if (q == 0 && lane == 0) sv[q].x = *h_word(s, node, HK_TI, tv.bw) & tv.mask;
      } else if (j < tv.nw) {
        const int x = tv.bw + j;
        if ((x >> 1) == (node >> 7)) {
          sv[q].x = *h_word(s, node, HK_TI, x);
          sv[q].y = *h_word(s, node, HK_TI, x + 1);
        } else {
          sv[q] = gld((const V2 WG_G*)h_word(s, node, HK_TI, x));
        }
      }
    }
    if (open) {
      const int fbw = sib_view(node, lane).bw;  // lane = level: the level's sibling block
      const uint64_t w1 = ld_coherent(h_dword(s, node, lane, (cand >> 6) - fbw) + HD_FP);
      const uint64_t w2 = two ? ld_coherent(h_dword(s, node, lane, (cand2 >> 6) - fbw) + HD_FP) : ~0ULL;
      fin = (w1 >> (cand & 63)) & 1ULL;
      fin2 = (w2 >> (cand2 & 63)) & 1ULL;
      if (ATK) {  // a blacklisted peer is passed over like a finished one (:493)
        fin = fin || blk(s, node, cand);
        if (two) fin2 = fin2 || blk(s, node, cand2);
      }
      if (fin && two && !fin2) {  // the first candidate is rejected (no effect but posInLevel++), the second one is taken
        cand = cand2;
        candT = cand2T;
        myPos++;
        fin = false;
      }
    }
    const bool lf = cti == mySize;  // incomingComplete :524-526
    {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int j = 2 * (q * 64 + lane);
        if (tv.nw == 1) {
          if (q == 0 && lane == 0) s.snap[refBase] = sv[q].x;
        } else if (j < tv.nw) {
          gst((V2 WG_G*)(s.snap + refBase + j), sv[q]);
        }
      }
      for (int j0 = 256; j0 < tv.nw; j0 += 256) {  // (beyond 16 384 ids per block: N > 32 768)
        uint64_t v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int j = j0 + q * 64 + lane;
          v[q] = j < tv.nw ? *h_word(s, node, HK_TI, tv.bw + j) & tv.mask : 0ULL;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int j = j0 + q * 64 + lane;
          if (j < tv.nw) s.snap[refBase + j] = v[q];
        }
      }
    }
    KPROF_MARK(c.d.g, 9);   // snapshots
    const uint64_t okM = __ballot(open && !fin);
    KPROF_MARK(c.d.g, 8);   // candidate peers' finished bits
    if (open && !fin) ls->pos[lane] = myPos + 1 >= mySize ? 0 : myPos + 1;  // getRemainingPeers(1) took the candidate
    __builtin_amdgcn_wave_barrier();
    // levels whose first candidate is a finished peer: the sequential scan of getRemainingPeers, before any send
    // (the scan does not depend on the sends; the records keep level order through their rank)
    uint64_t sendM = okM;
    for (uint64_t m = openM & ~okM; m; m &= m - 1) {
      const int l = __ffsll((unsigned long long)m) - 1;
      uint32_t dest = 0;
      const int got = remaining_peers(c, s, ls, l, 1, 0xFFFFFFFFu, &dest);
      KPROF_COUNT(c.d.g, 13);
      if (got <= 0) continue;
      if (lane == l) {
        candT = dest;
        cand = h_tag_id(s, dest);
      }
      sendM |= 1ULL << l;
    }
    if (sendM) {  // one rd.nextInt() per send, in level order (:374-382): lane l writes level l's record
      long long bytes = 0;
      for (uint64_t m = sendM; m; m &= m - 1) bytes += h_msg_size(__ffsll((unsigned long long)m) - 1);
      c.send_many((sendM >> lane) & 1ULL, __popcll(sendM & lanes_lt()), __popcll(sendM), cand,
                  (uint32_t)lane | (lf ? 32u : 0u) | h_msg_rank(s, candT),
                  complete ? H_REF_ONES : refBase + (uint32_t)(own_view(node, lane >= 1 ? lane : 1).bw - tv.bw), bytes);
    }
    if (s.xS > 1 && sendM) {
      // owner-directed exchange (HandelState::xout): the sub-rows of this snapshot that messages to OTHER shards' nodes point at
      const bool cross = ((sendM >> lane) & 1ULL) && !complete && h_owner(s, cand) != s.xMe;
      const uint64_t xm = __ballot(cross);
      if (xm) __threadfence_block();  // (the words this wavefront stored into the ring above are read back through L2 below)
      for (uint64_t m = xm; m; m &= m - 1) {
        const int lv = __ffsll((unsigned long long)m) - 1;
        const int dst = h_owner(s, (int32_t)lane_bcast((uint32_t)cand, lv));
        const Lv ov = own_view(node, lv);
        const uint32_t ref = refBase + (uint32_t)(ov.bw - tv.bw);
        const int nch = (ov.nw + H_XWORDS - 1) / H_XWORDS;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(F(s.xoutCount + dst), (uint32_t)nch);
        base = lane_bcast(base, 0);
        if (base + (uint32_t)nch > s.xChunkCap) {
          if (lane == 0) set_err(c.d.g, ERR_PAYLOAD);
          continue;
        }
        for (int c0 = 0; c0 < nch; c0 += 64 / H_XWORDS) {  // four chunks a round: lane = chunk * 16 + word
          const int ch = c0 + (lane >> 4), j = lane & (H_XWORDS - 1), wi = ch * H_XWORDS + j;
          if (ch < nch) {
            uint64_t WG_G* o = s.xout + ((size_t)dst * s.xChunkCap + base + (uint32_t)ch) * H_XCHUNK;
            if (wi < ov.nw) o[1 + j] = ld_coherent(s.snap + ref + wi);
            if (j == 0) o[0] = (uint64_t)(ref + (uint32_t)(ch * H_XWORDS)) | ((uint64_t)min(H_XWORDS, ov.nw - ch * H_XWORDS) << 32);
          }
        }
      }
    }
    KPROF_MARK(c.d.g, 10);  // sends
  }

  // the fast path of updateVerifiedSignatures (:738-749) for the levels above `lv` that just got a complete
  // totalOutgoing: getRemainingPeers(fastPath) + one multi-destination send each. Run right after the update by the
  // wave-per-node visit, or later in the same pass for an update a lane applied (k_handel_lane defers it: the scan of the
  // emission list is wave-parallel work) — nothing between the two moments touches what it reads (the node's other
  // events of the ms are SendSigs deliveries: they change neither |totalIncoming| nor posInLevel / outgoingFinished).
  __device__ static void fast_path(Ctx& c, const State& s, LevelScalars* ls, int lv) {
    const int lane = WG_LANE;
    // totalOutgoing(l) = the sum of |totalIncoming| below l (:728-731): lane l takes level l, one prefix scan over the lanes
    const int myCTI = lane < s.L ? ls->cTI[lane] : 0;
    const int incl = (int)wave_incl_scan32((uint32_t)myCTI);
    const bool fp = s.p.fastPath > 0 && lane > lv && lane < s.L && !ls->outFin[lane] && incl - myCTI == (1 << (lane - 1));
    for (uint64_t fpM = __ballot(fp); fpM; fpM &= fpM - 1) {
      const int l = __ffsll((unsigned long long)fpM) - 1;
      __threadfence_block();  // the snapshot below reads the totalIncoming words stored above
      uint32_t destOff = c.dest_reserve(s.p.fastPath);
      int n = remaining_peers(c, s, ls, l, s.p.fastPath, destOff, nullptr);
      if (n > 0) {
        uint32_t ref = snapshot_outgoing(c, s, l);
        bool lf = ls->cTI[l] == (1 << (l - 1));
        __threadfence_block();
        c.send_list(destOff, n, (uint32_t)l | (lf ? 32u : 0u), ref, h_msg_size(l));
      }
    }
  }

  // ---- Task: updateVerifiedSignatures (:690-754), one wavefront per node ----------------------------------
  __device__ static void update_verified(Ctx& c, const State& s, NodeRegs& r, uint32_t arg) {
    const int32_t node = c.node;
    const int lane = WG_LANE;
    LevelScalars* ls = r.ls;
    const int pk = H_ARG_PK(arg), lv = H_ARG_LV(arg), slot = H_ARG_SLOT(arg);
    const int32_t from = H_ARG_FROM(arg);
    const uint32_t pe = WG_READFIRST(ls->sc[HH_PEND + pk]);
    if (pe != h_pend_word(lv, slot) || (int32_t)WG_READFIRST(ls->sc[HH_PENDFROM + pk]) != from) {
      if (lane == 0) set_err(c.d.g, ERR_PROTOCOL);
      return;
    }
    const int total0 = (int)WG_READFIRST(ls->sc[HH_TOTAL]);  // the sum of |totalIncoming| over the levels, before this task
    __builtin_amdgcn_wave_barrier();  // every lane has read the record before lane 0 clears it
    if (lane == 0) ls->sc[HH_PEND + pk] = 0;
    if (ATK && s.atk == 1) {  // :688-694 a bad signature: its signer is blacklisted, nothing else happens (the entry stays listed)
      const uint64_t badM = h_qrec(s, node, lv)[s.qBad];
      if ((badM >> slot) & 1ULL) {
        if (lane == 0) atomicOr((unsigned long long*)(s.blacklist + (size_t)node * s.W + (from >> 6)), 1ULL << (from & 63));
        __builtin_amdgcn_wave_barrier();
        return;
      }
    }
    const Lv v = sib_view(node, lv);
    uint64_t WG_G* ti = h_row(s, node, HK_TI, lv);  // the level's block: word j of it is [j]
    uint64_t WG_G* la = h_row(s, node, HK_LA, lv);
    uint64_t WG_G* vi = h_row(s, node, HK_VI, lv);
    const uint64_t WG_G* sig = sig_ptr(s, node, lv, slot);
    // ---- every load of the event, issued before the first use (one memory round trip)
    const int jF = (from >> 6) - v.bw;  // `from` lies in the level's block
    const uint64_t bit = 1ULL << (from & 63);
    // (toVerifyInd.set(from, false) :701 — the set is SEEN & ~verifiedInd, and verifiedInd gets `from` below: nothing to store)
    uint64_t WG_G* qr = h_qrec(s, node, lv);
    uint64_t WG_G* ent = qr + H_QENT;
    const HQHead qh = gld((const HQHead WG_G*)qr);
    uint64_t entAll = lane < 13 ? ent[lane] : 0ULL;  // (the record's first two lines, before the list's length is known)
    const int j0 = lane;
    const bool has0 = j0 < v.nw;
    uint64_t sg0 = 0, vi0 = 0, la0 = 0, ti0 = 0;
    if (has0) {
      sg0 = sig[j0];
      vi0 = vi[j0];
      la0 = la[j0];
      ti0 = la0 | vi0;  // totalIncoming = lastAggVerified | verifiedIndSignatures: the row is written, never read (file header)
    }
    // the VI / TI words holding `from` are among the block words just loaded (the lane owning block word jF);
    // only beyond the first 64 words of a wide level do they cost memory instructions of their own
    uint64_t viF, tiF;
    if (jF < 64) {
      viF = lane_bcast64(vi0, jF);
      tiF = lane_bcast64(ti0, jF);
    } else {
      viF = ld_coherent(vi + jF);
      tiF = viF | ld_coherent(la + jF);
    }
    const bool owner = lane == (jF & 63);
    // toVerifyAgg.remove(vs): identity remove, sigQueueSize untouched
    const int len = (int)qh.len;
    if (len > 13) entAll = ent[lane];
    const uint64_t myEnt = lane < len ? entAll : ~0ULL;
    int newLen = len;
    {
      const uint64_t hit = __ballot(lane < len && (int)(myEnt & 0xFF) == slot);
      if (hit) {
        const int at = __ffsll((unsigned long long)hit) - 1;
        const uint64_t next = shfl64(myEnt, (lane + 1) & 63);
        if (lane >= at && lane < len - 1) ent[lane] = next;
        newLen = len - 1;
        if (lane == 0 && len == 1) ls->sc[HH_QMASK] &= ~(1u << lv);
      }
    }
    const bool hadVI = (viF & bit) != 0, hadTI = (tiF & bit) != 0;
    // verifiedIndSignatures.set(from); totalIncoming.set(from) if new — applied to the register copies of
    // the words and written back by the owning lane
    if (owner) {
      if (!hadVI) vi[jF] = viF | bit;
    }
    int cVI = ls->cVI[lv] + (hadVI ? 0 : 1);
    const int cTI0 = ls->cTI[lv];
    int cTI = cTI0;
    int cLA = ls->cLA[lv];
    bool improved = false;
    if (!hadTI) {
      cTI++;
      improved = true;
    }
    const uint64_t ti0m = ti0;  // the word as it is in memory
    if (has0 && j0 == jF) {
      vi0 |= bit;
      if (!hadTI) ti0 |= bit;
    }
    // all = sig | verifiedInd ; intersects(lastAgg, sig)
    uint64_t acc = 0;
    if (has0) acc = (uint64_t)__popcll(sg0 | (vi0 & v.mask)) | ((uint64_t)((sg0 & la0 & v.mask) != 0) << 32);
    for (int j = j0 + 64; j < v.nw; j += 64) {
      uint64_t sg = sig[j], viw = vi[j] & v.mask, law = la[j] & v.mask;
      if (j == jF) viw |= bit;
      acc += (uint64_t)__popcll(sg | viw) | ((uint64_t)((sg & law) != 0) << 32);
    }
    acc = wave_sum64(acc);
    const int u2 = (int)(acc & 0xFFFFFFFFu);
    const bool inter = (acc >> 32) != 0;
    if (u2 > cVI) {
      improved = true;
      uint64_t cnt = 0;
      // (only the words that change are written: a verified aggregate usually adds a few bits to a block whose other
      // words — up to 2 KB of them — would otherwise be dirtied in L2 and written back to HBM for nothing)
      if (has0) {
        uint64_t nla = (inter ? 0ULL : (la0 & v.mask)) | sg0;
        uint64_t nti = nla | (vi0 & v.mask);
        if (nla != (la0 & v.mask)) la[j0] = (la0 & ~v.mask) | nla;
        if (nti != (ti0m & v.mask)) ti[j0] = (ti0m & ~v.mask) | nti;
        cnt = (uint64_t)__popcll(nla) | ((uint64_t)__popcll(nti) << 32);
      }
      for (int j = j0 + 64; j < v.nw; j += 64) {
        uint64_t sg = sig[j];
        uint64_t law = la[j], viw = vi[j];
        const uint64_t tiw = law | (j == jF ? (viw & ~(hadVI ? 0ULL : bit)) : viw);  // the TI word as it is in memory
        if (j == jF) viw |= bit;  // (this lane stored it above; same-lane order makes the reload see it anyway)
        uint64_t nla = (inter ? 0ULL : (law & v.mask)) | sg;
        uint64_t nti = nla | (viw & v.mask);
        if (nla != (law & v.mask)) la[j] = (law & ~v.mask) | nla;
        if (nti != (tiw & v.mask)) ti[j] = (tiw & ~v.mask) | nti;
        cnt += (uint64_t)__popcll(nla) | ((uint64_t)__popcll(nti) << 32);
      }
      cnt = wave_sum64(cnt);
      cLA = (int)(cnt & 0xFFFFFFFFu);
      cTI = (int)(cnt >> 32);
    } else if (!hadTI && owner) {
      ti[jF] = tiF | bit;
    }
    const int cur = total0 + (cTI - cTI0);  // `cur.cardinality()` of :745 after the loop over the levels
    if (lane == 0) {
      ls->cVI[lv] = cVI;
      ls->cTI[lv] = cTI;
      ls->cLA[lv] = cLA;
      ls->sc[HH_TOTAL] = (uint32_t)cur;
      // The entry was just unlisted (an entry is listed at most once), so its slot dies with this task
      // unless another registered task still references it (checkSigs can pick the same entry twice).
      HQHead nh;
      nh.len = (uint64_t)newLen;
      nh.used = slot_pending(r, lv, slot) ? qh.used : (qh.used & ~(1ULL << slot));
      if (nh.len != qh.len || nh.used != qh.used) gst((HQHead WG_G*)qr, nh);
      if (improved || !hadVI) qr[H_QVALID] = 0;  // one of the level's three sets changed: every cached evaluation of its entries is void
      if (newLen > 0) ls->sc[HH_QDIRTY] |= 1u << lv;  // (the list lost an entry: the level's summary is stale either way)
    }
    __builtin_amdgcn_wave_barrier();
    if (!improved) return;
    if (cTI == v.size) fast_path(c, s, ls, lv);  // justCompleted = incomingComplete()
    if (r.doneAt == 0 && cur >= s.p.threshold) r.doneAt = c.t;
  }
};
typedef HandelProtoT<false> HandelProto;

// ------------------------------------------------------------------------------------------------
// The delivery pass, first kernel: one LANE per node with events (the reference applies an envelope to its `to` node,
// C/Network.java:594-635; a lane applies its node's <= 4 events of the ms in event order).
struct HLaneNode {
  long long doneAt, doneAt0;
  int32_t startAt, sigQueueSize, sigQueueSize0, msgFiltered, msgFiltered0;
  uint32_t qmask, qmask0;
  uint32_t qdirty;  // levels that got an entry without a cached evaluation (HH_QDIRTY), OR-ed into the header at the end
  int32_t total, total0;
};
// work descriptor of the wave-per-node kernel (16 bytes): a node visit {node, vflags << 8, events, events the lane kernel
// applied already} or a fast-path item {node, 1 | vflags << 8, event, level} the lane kernel deferred
constexpr uint32_t HW_VISIT = 0u, HW_FASTPATH = 1u;

// onNewSig (:757-790) by one lane; a payload wider than one word is handed back as a copy job
__device__ __forceinline__ void h_lane_message(const EngineDev& d, const HandelState& s, int32_t t, int32_t node, HLaneNode& r,
                                               int32_t from, uint32_t msg, uint32_t payload, CopyJob& job) {
  const int l = (int)(msg & 31u);
  const bool levelFinished = (msg >> 5) & 1u;
  if (r.doneAt > 0) {  // :758-761
    r.msgFiltered++;
    return;
  }
  if (t < r.startAt) return;
  const int w = (from >> 6) - sib_view(node, l).bw;  // the word of `from` inside the level's block
  const uint64_t bit = 1ULL << (from & 63);
  uint64_t WG_G* dp = h_dword(s, node, l, w);  // the {SEEN, FP, BUMP} words holding `from`: one 32-byte piece
  // every load of the event before the first use: the piece's words (BUMP: CARRIED ranks only) or the matrix entry, the record's head
  V2 sf;  // {SEEN, FP} (pieces are 24 bytes apart: 8-byte loads)
  sf.x = dp[HD_SEEN];
  sf.y = dp[HD_FP];
  const uint64_t bumpW = s.ranks ? 0ULL : dp[HD_BUMP];
  const int32_t rankM = s.ranks ? s.ranks[(size_t)node * s.N + from] : 0;
  uint64_t WG_G* qr = h_qrec(s, node, l);
  const HQHead qh = gld((const HQHead WG_G*)qr);
  const uint64_t qvalid = qr[H_QVALID];
  const unsigned long long used = qh.used;
  const int len = (int)qh.len;
  const uint64_t WG_G* src = h_payload(d, s, payload);
  const int nw = h_nw(l);
  const uint64_t pw0 = nw == 1 ? src[0] & sib_view(node, l).mask : 0ULL;
  // finishedPeers.set(from) if levelFinished; toVerifyInd.set(from) unless verified = SEEN & ~VI (file header). Plain stores of
  // the loaded words: this lane is the node's only writer in its launch, and an L2 atomic costs several stores
  // (profiles/r20e: three atomics a node were 22 of the kernel's 92 us)
  if (!(sf.x & bit)) dp[HD_SEEN] = sf.x | bit;
  if (levelFinished && !(sf.y & bit)) dp[HD_FP] = sf.y | bit;
  const int32_t rank = s.ranks ? rankM : h_rank_at_delivery(s, node, from, msg, bumpW);  // receptionRanks[from], read at receive time (:784)
  r.sigQueueSize++;
  const int qc = h_qcap(s, l);
  const unsigned long long capMask = qc >= 64 ? ~0ULL : ((1ULL << qc) - 1ULL);
  const unsigned long long freeM = ~used & capMask;
  if (freeM == 0 || len >= 64) {
    set_err(d.g, ERR_QUEUE_CAP);
    return;
  }
  const int slot = __ffsll(freeM) - 1;
  uint64_t WG_G* dst = h_sig_ptr(s, node, l, slot);
  qr[H_QENT + len] = h_entry(rank, from, slot);  // (lists of up to five entries: the same line as the head)
  HQHead nh;
  nh.len = (uint64_t)(len + 1);
  nh.used = used | (1ULL << slot);
  gst((HQHead WG_G*)qr, nh);
  if ((qvalid >> slot) & 1ULL) qr[H_QVALID] = qvalid & ~(1ULL << slot);  // the slot's cached evaluation was its previous entry's
  r.qmask |= 1u << l;
  r.qdirty |= 1u << l;
  if (nw == 1) {
    dst[0] = pw0;
  } else {
    job.src = src;
    job.dst = dst;
    job.nw = nw;
  }
}

// updateVerifiedSignatures (:690-754) of a level whose block is <= H_LANE_NW words, by one lane.
//   H_UPD_DONE   applied, nothing follows
//   H_UPD_DEFER  applied; the level just became complete and a fast path may follow (:738-749): the caller hands that
//                part to k_handel_wave. Only when the task is the node's LAST event of the ms: the scan of the emission
//                lists reads finishedPeers, which a later SendSigs of the same ms would have changed by then
//   H_UPD_BAIL   nothing applied: the level would become complete and the node has later events — the rest of the visit
//                (from this event on) goes to k_handel_wave, which runs the fast path in its place
enum : int { H_UPD_DONE = 0, H_UPD_DEFER = 1, H_UPD_BAIL = 2 };
__device__ __forceinline__ int h_lane_update(const EngineDev& d, const HandelState& s, int32_t t, int32_t node, HLaneNode& r,
                                             uint32_t arg, bool hasLater) {
  const int pk = H_ARG_PK(arg), lv = H_ARG_LV(arg), slot = H_ARG_SLOT(arg);
  const int32_t from = H_ARG_FROM(arg);
  uint32_t WG_G* hdr = h_hdr(s, node);
  const Lv v = sib_view(node, lv);
  uint64_t WG_G* ti = h_row(s, node, HK_TI, lv);  // (the level's five bitsets side by side)
  uint64_t WG_G* la = h_row(s, node, HK_LA, lv);
  uint64_t WG_G* vi = h_row(s, node, HK_VI, lv);
  const uint64_t WG_G* sig = h_sig_ptr(s, node, lv, slot);
  const int jF = (from >> 6) - v.bw;  // `from` lies in the level's block: 0 <= jF < nw
  const uint64_t bit = 1ULL << (from & 63);
  uint64_t WG_G* qr = h_qrec(s, node, lv);
  uint64_t WG_G* ent = qr + H_QENT;
  // ---- the loads that depend on the task's argument only, before the first use
  const U4 pend = gld((const U4 WG_G*)(hdr + HH_PEND));
  const U4 pfrom = gld((const U4 WG_G*)(hdr + HH_PENDFROM));
  U4 WG_G* lvA = (U4 WG_G*)h_lv(s, node, HP_POS, lv);   // {posInLevel, |totalIncoming|, |lastAggVerified|, |verifiedInd|}
  U4 a = gld(lvA);
  const HQHead qh = gld((const HQHead WG_G*)qr);
  const uint64_t viF = vi[jF], tiF = viF | la[jF];  // totalIncoming = lastAggVerified | verifiedInd: the row is not read
  uint64_t e6[6];  // the head of the level's list (the record's first line); longer lists are walked in memory below
#pragma unroll
  for (int i = 0; i < 6; i++) e6[i] = ent[i];
  // ---- the pending-table entry this task owns
  const uint32_t pe = pk == 0 ? pend.x : pk == 1 ? pend.y : pk == 2 ? pend.z : pend.w;
  const uint32_t pf = pk == 0 ? pfrom.x : pk == 1 ? pfrom.y : pk == 2 ? pfrom.z : pfrom.w;
  if (pe != h_pend_word(lv, slot) || (int32_t)pf != from) {
    set_err(d.g, ERR_PROTOCOL);
    return H_UPD_DONE;
  }
  // ---- what the task will do to the level's sets, before anything is stored (H_UPD_BAIL leaves the node untouched)
  const bool hadVI = (viF & bit) != 0, hadTI = (tiF & bit) != 0;
  int cVI = (int)a.w + (hadVI ? 0 : 1);
  const int cTI0 = (int)a.y;
  int cTI = cTI0, cLA = (int)a.z;
  bool improved = false;
  if (!hadTI) {
    cTI++;
    improved = true;
  }
  // all = sig | verifiedInd (with `from`) ; intersects(lastAgg, sig)
  int u2 = 0;
  bool inter = false;
  h_stream4(sig, vi, la, la, v.nw, [&](int j, uint64_t sg, uint64_t viw, uint64_t law, uint64_t) {
    viw |= j == jF ? bit : 0ULL;
    u2 += __popcll(sg | (viw & v.mask));
    inter |= (sg & law & v.mask) != 0;
  });
  const bool replace = u2 > cVI;  // all.cardinality() > verifiedIndSignatures.cardinality(): the aggregate replaces / extends lastAggVerified
  const bool mayComplete = s.p.fastPath > 0 && lv + 1 < s.L;  // (a complete level lets the levels above take the fast path)
  if (replace) {
    improved = true;
    if (hasLater && mayComplete) {  // would the level become complete? (counted before anything is stored)
      int nTI = 0;
      h_stream4(sig, vi, la, la, v.nw, [&](int j, uint64_t sg, uint64_t viw, uint64_t law, uint64_t) {
        viw |= j == jF ? bit : 0ULL;
        nTI += __popcll((inter ? 0ULL : (law & v.mask)) | sg | (viw & v.mask));
      });
      if (nTI == v.size) return H_UPD_BAIL;
    }
  } else if (hasLater && mayComplete && improved && cTI == v.size) {
    return H_UPD_BAIL;
  }
  // ---- apply
  hdr[HH_PEND + pk] = 0;
  // (toVerifyInd.set(from, false) :701: the set is SEEN & ~verifiedInd — nothing to store)
  // toVerifyAgg.remove(vs): identity remove, sigQueueSize untouched
  const int len = (int)qh.len;
  HQHead nh = qh;
  {
    // (static indices throughout: a register array indexed at run time would live in scratch memory)
    int at = -1;
#pragma unroll
    for (int i = 0; i < 6; i++)
      if (i < len && at < 0 && (int)(e6[i] & 0xFF) == slot) at = i;
    for (int i = 6; i < len && at < 0; i++)
      if ((int)(ent[i] & 0xFF) == slot) at = i;
    if (at >= 0) {
#pragma unroll
      for (int i = 0; i < 5; i++)
        if (i >= at && i + 1 < len) ent[i] = e6[i + 1];
      for (int i = at > 5 ? at : 5; i + 1 < len; i++) ent[i] = ent[i + 1];
      nh.len = (uint64_t)(len - 1);
      if (len == 1) r.qmask &= ~(1u << lv);
    }
  }
  if (replace) {  // (only the words that change are written)
    cLA = 0;
    cTI = 0;
    h_stream4(sig, vi, la, la, v.nw, [&](int j, uint64_t sg, uint64_t viw, uint64_t law, uint64_t tiw) {
      tiw = law | viw;  // the TI word as it is in memory
      const uint64_t viN = viw | (j == jF ? bit : 0ULL);
      const uint64_t nla = (inter ? 0ULL : (law & v.mask)) | sg;
      const uint64_t nti = nla | (viN & v.mask);
      if (nla != (law & v.mask)) la[j] = (law & ~v.mask) | nla;
      if (nti != (tiw & v.mask)) ti[j] = (tiw & ~v.mask) | nti;
      cLA += __popcll(nla);
      cTI += __popcll(nti);
    });
  } else if (!hadTI) {
    ti[jF] = tiF | bit;
  }
  if (!hadVI) vi[jF] = viF | bit;
  if (improved || !hadVI) qr[H_QVALID] = 0;  // one of the level's three sets changed: every cached evaluation of its entries is void
  if (nh.len > 0) r.qdirty |= 1u << lv;       // (the list lost an entry: the level's summary is stale either way)
  a.y = (uint32_t)cTI;
  a.z = (uint32_t)cLA;
  a.w = (uint32_t)cVI;
  gst(lvA, a);
  // The entry was just unlisted (an entry is listed at most once), so its slot dies with this task unless another
  // registered task still references it (checkSigs can pick the same entry twice).
  const uint32_t key = h_pend_word(lv, slot);
  const bool held = (pk != 0 && pend.x == key) || (pk != 1 && pend.y == key) || (pk != 2 && pend.z == key) || (pk != 3 && pend.w == key);
  if (!held) nh.used &= ~(1ULL << slot);
  if (nh.len != qh.len || nh.used != qh.used) gst((HQHead WG_G*)qr, nh);
  r.total += cTI - cTI0;
  if (!improved) return H_UPD_DONE;
  if (r.doneAt == 0 && r.total >= s.p.threshold) r.doneAt = t;
  return (cTI == v.size && mayComplete) ? H_UPD_DEFER : H_UPD_DONE;  // justCompleted
}

// A lane's visit ends: what it changed of the header's first line goes back as the 16-byte pieces it loaded (h0 = words
// 0..3, h2 = 8..11, h3 = 12..15) — the lane is the node's only writer in its launch, so the counters (Node.msgReceived /
// bytesReceived, C/Network.java:611-612) and the dirty-level mask are plain stores of loaded value + delta, not three
// atomics a node (565 k L2 atomics per launch at 24 copies: profiles/r20d_pmc_req.md)
__device__ __forceinline__ void h_lane_store_header(uint32_t WG_G* hdr, const HLaneNode& r, const U4& h0, const U4& h2, const U4& h3,
                                                    long long nRecv, long long bRecv) {
  if (r.sigQueueSize != r.sigQueueSize0 || r.msgFiltered != r.msgFiltered0) {
    U4 n0 = h0;
    n0.y = (uint32_t)r.sigQueueSize;
    n0.z = (uint32_t)r.msgFiltered;
    gst((U4 WG_G*)hdr, n0);
  }
  if (nRecv || r.qmask != r.qmask0 || r.doneAt != r.doneAt0) {
    U4 n2;
    n2.x = h2.x + (uint32_t)nRecv;
    n2.y = (uint32_t)(unsigned long long)r.doneAt;
    n2.z = (uint32_t)((unsigned long long)r.doneAt >> 32);
    n2.w = r.qmask;
    gst((U4 WG_G*)(hdr + 8), n2);
  }
  if (nRecv || r.qdirty) {
    const unsigned long long br = ((unsigned long long)h3.x | ((unsigned long long)h3.y << 32)) + (unsigned long long)bRecv;
    U4 n3 = h3;
    n3.x = (uint32_t)br;
    n3.y = (uint32_t)(br >> 32);
    n3.w |= r.qdirty;
    gst((U4 WG_G*)(hdr + 12), n3);
  }
}

// (three wavefronts a SIMD: 168 registers without a spill. Left to itself the compiler takes 172 — and the kernel drops to two:
// 84 -> 81 us per launch at 31 copies, profiles/r22h_*)
__global__ void __launch_bounds__(256, 3) k_handel_lane(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  const int lane = WG_LANE;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t nActive = d.g->nActive;
  const int32_t t = d.g->now;
  U4 WG_G* work = (U4 WG_G*)(VisitDesc WG_G*)d.activeB;  // the wave-per-node kernel's list
  for (uint32_t base = wave * 64; base < nActive; base += nWaves * 64) {
    KPROF_DECL;
    KPROF_COUNT(d.g, 24);
    const uint32_t a = base + lane;
    const bool have = a < nActive;
    const int32_t node = have ? (int32_t)d.active[a] : 0;
    uint32_t cnt = 0, vflags = 0;
    InboxEntry E[INBOX_SLOTS];
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) E[k].e = 0xFFFFFFFFu, E[k].w0 = 0, E[k].w2 = 0, E[k].w3 = 0;
    // header words 0..3 {addedCycle, sigQueueSize, msgFiltered, startAt}, 8..11 {msgReceived, doneAt lo, hi, queue mask} and
    // 12..15 {bytesReceived lo, hi, msgSent, dirty levels}: pieces of ONE line, written back as pieces (h_lane_store_header)
    U4 h0, h2;
    h0.x = h0.y = h0.z = h0.w = 0;
    h2 = h0;
    uint32_t total = 0;
    if (have) {
      cnt = d.icnt[node];
      d.icnt[node] = 0;  // the line is consumed by this pass (its entries stay readable for the wave-per-node kernel)
#pragma unroll
      for (int k = 0; k < INBOX_SLOTS; k++) E[k] = gld(d.inbox + ((size_t)node * INBOX_SLOTS + k));
      vflags = (d.nodes.down[node] ? VD_DOWN : 0u) | (d.nparts ? (uint32_t)d.nodes.part[node] << 8 : 0u);
      const uint32_t WG_G* hdr = h_hdr(s, node);
      h0 = gld((const U4 WG_G*)hdr);
      h2 = gld((const U4 WG_G*)(hdr + 8));
      total = gld((const U4 WG_G*)(hdr + 4)).w;  // HH_TOTAL (the four pieces are one 64-byte line)
    }
    // the node's events in event order (the line is in arrival order of the expand lanes)
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++)
      if ((uint32_t)k >= cnt) E[k].e = 0xFFFFFFFFu;
#define H_CSWAP(A, B)                  \
  if (E[B].e < E[A].e) {               \
    const InboxEntry x = E[A];         \
    E[A] = E[B];                       \
    E[B] = x;                          \
  }
    H_CSWAP(0, 1) H_CSWAP(2, 3) H_CSWAP(0, 2) H_CSWAP(1, 3) H_CSWAP(1, 2)
#undef H_CSWAP
    // ---- which kernel applies the node's events: this lane, if they are <= 4 SendSigs deliveries and at most one
    // updateVerifiedSignatures — of a narrow level (applied here), or of a wide level as the node's LAST event (handed to
    // k_handel_update: the deliveries before it are this lane's); else a wavefront of k_handel_wave
    bool mine = have && cnt <= (uint32_t)INBOX_SLOTS && !s.atk;  // (byzantineSuicide: every visit by k_handel_wave)
    // the node's first event is its dissemination: the lean kernel takes that event, k_handel_wave the rest
    const bool disFirst = mine && ((E[0].w0 >> 28) & 3u) == K_PERIODIC && E[0].w2 == H_TASK_DISSEMINATION;
    if (disFirst) mine = false;
    {
      const uint64_t m = __ballot(disFirst);
      if (m) {
        uint32_t bb = 0;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if (lane == leader) bb = atomicAdd(F(s.disCount + 0), (uint32_t)__popcll(m));
        bb = lane_bcast(bb, leader);
        if (disFirst) {
          U4 q;
          q.x = (uint32_t)node;
          q.y = (vflags << 8) | cnt;
          q.z = E[0].e;
          q.w = E[0].w0;
          gst((U4 WG_G*)s.itemsDis + (bb + __popcll(m & lanes_lt())), q);
        }
      }
    }
    int nUpd = 0, wideAt = -1;
    // A wide update as the node's FIRST event with only plain SendSigs deliveries behind it (the usual shape: the task was
    // registered a pairing time ago, most messages of the ms were sent before that and so come after it in the LIFO's
    // order) is k_handel_update's too since round 4: that kernel applies the update and then the deliveries, in event
    // order — this lane applies nothing of the node. (Until then such a node was a visit of k_handel_wave: a third of that
    // kernel's items.)
    bool updFirst = false, plainBehind = true;
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) {
      if ((uint32_t)k < cnt) {
        const uint32_t kind = (E[k].w0 >> 28) & 3u;
        if (k > 0 && (kind != K_MSG || (E[k].w0 & INBOX_CHAIN))) plainBehind = false;
        if (kind == K_MSG) {
          // (a hop of a fast-path envelope too: its re-push after the run's last hop is one record, see below)
        } else if (kind == K_TASK && E[k].w2 == H_TASK_UPDATE) {
          nUpd++;
          const int unw = h_nw(H_ARG_LV(E[k].w3));
          if (unw > s.laneNw) {
            if ((uint32_t)(k + 1) == cnt && unw <= H_UPD_NW)
              wideAt = k;
            else {
              mine = false;
              updFirst = k == 0 && unw <= H_UPD_NW;
            }
          }
        } else {
          mine = false;
        }
      }
    }
    if (nUpd > 1) mine = false;
    updFirst = updFirst && have && plainBehind && nUpd == 1 && cnt <= (uint32_t)INBOX_SLOTS && !s.atk && !disFirst;
    // the dissemination first and only plain deliveries behind it: those are a lane's (k_handel_lane2's second launch, behind k_handel_dissem)
    const bool disTrail = disFirst && cnt > 1u && plainBehind && nUpd == 0 && !s.atk;
    {
      const uint64_t tm = __ballot(disTrail);
      if (tm) {
        uint32_t tb = 0;
        const int leader = __ffsll((unsigned long long)tm) - 1;
        if (lane == leader) tb = atomicAdd(F(s.trail2Count + 0), (uint32_t)__popcll(tm));
        tb = lane_bcast(tb, leader);
        if (disTrail) {
          U4 q;
          q.x = (uint32_t)node;
          q.y = (vflags << 8) | cnt;
          q.z = E[0].e;  // (the dissemination: not this list's)
          q.w = 0;
          gst((U4 WG_G*)s.itemsTrail2 + (tb + __popcll(tm & lanes_lt())), q);
        }
      }
    }
    {  // the rest goes to the wave-per-node kernel: one atomic per wavefront
      // (a node whose first event k_handel_dissem applies: the rest of its events, if any, as a visit that skips the first —
      // listed here, with this wavefront's one atomic, not by k_handel_dissem with one atomic per node on the same word)
      const bool toB = have && !mine && !updFirst && !disTrail && (!disFirst || cnt > 1u);
      const uint64_t m = __ballot(toB);
      if (m) {
        uint32_t bb = 0;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if (lane == leader) bb = atomicAdd(F(&d.g->nActiveB), (uint32_t)__popcll(m));
        bb = lane_bcast(bb, leader);
        if (toB) {
          U4 q;
          q.x = (uint32_t)node;
          q.y = HW_VISIT | (vflags << 8);
          q.z = cnt;
          q.w = disFirst ? 1u : 0u;  // (events to skip)
          gst(work + (bb + __popcll(m & lanes_lt())), q);
        }
      }
    }
    HLaneNode r;
    r.doneAt = r.doneAt0 = (long long)((unsigned long long)h2.y | ((unsigned long long)h2.z << 32));
    r.startAt = (int32_t)h0.w;
    r.sigQueueSize = r.sigQueueSize0 = (int32_t)h0.y;
    r.msgFiltered = r.msgFiltered0 = (int32_t)h0.z;
    r.qmask = r.qmask0 = h2.w;
    r.qdirty = 0;
    r.total = r.total0 = (int32_t)total;
    const bool toDown = (vflags & VD_DOWN) != 0;
    const uint8_t toPart = (uint8_t)(vflags >> 8);
    KPROF_MARK(d.g, 25);  // inbox line + header + classification
    long long nRecv = 0, bRecv = 0;
    bool fpDefer = false;            // the node hands k_handel_wave an item: a deferred fast path, or (bailAt >= 0) the
    uint32_t fpEvent = 0, fpLevel = 0;  // rest of its visit from event bailAt on
    int bailAt = -1;
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) {
      CopyJob job;
      job.nw = 0;
      if (mine && bailAt < 0 && E[k].e != 0xFFFFFFFFu) {
        const uint32_t e = E[k].e;
        EvRes res;
        res.nrec = 0;
        res.ndraw = 0;
        bool deferred = false;
        if (((E[k].w0 >> 28) & 3u) == K_MSG) {
          const int32_t from = (int32_t)(E[k].w0 & 0x0FFFFFFFu);
          const bool hop = (E[k].w0 & INBOX_CHAIN) != 0;
          EvAux ax;
          ax.chain = -1;
          ax.cpos = 0;
          ax.outBase = ax.outCap = 0;
          if (hop) ax = gld(d.evAux + e);  // (requested before the delivery's own loads are waited for)
          if (!toDown && (d.nparts == 0 || d.nodes.part[from] == toPart)) {  // C/Network.java:606
            nRecv++;
            bRecv += h_msg_size((int)(E[k].w2 & 31u));
            res.nrec = EV_DELIVERED | ((E[k].w2 & 31u) << 24);
            h_lane_message(d, s, t, node, r, from, E[k].w2, E[k].w3, job);
          }
          if (hop && ax.cpos < 0) {  // last hop of the run: markRead(); if (hasNextReader()) msgs.addMsg(m)  :629-632
            const int32_t next = (ax.cpos & 0x7FFFFFFF) + 1;
            if (next < d.chains[ax.chain].ndest) {
              if (ax.outCap > 0 && ax.outBase < d.maxOut) {
                Out o;  // (what Ctx::put writes for O_CHAINCONT: the event's one record)
                o.kindfrom = (O_CHAINCONT << 28) | (uint32_t)node;
                o.to = ax.chain;
                o.a = (uint32_t)next;
                o.b = 0;
                o.t = 0;
                o.destOff = 0;
                o.drawsub = 0;
                o.pad = 0;
                gst(d.outTmp + ax.outBase, o);
              } else {
                set_err(d.g, ERR_OUTBOX);
              }
              res.nrec |= 1u;
            } else {
              d.chains[ax.chain].flags = 0;  // envelope fully delivered
            }
          }
        } else if (!toDown && k == wideAt) {
          deferred = true;  // k_handel_update applies it (and writes the event's result)
        } else if (!toDown) {
          res.nrec = EV_TASK_RUN;
          const int st = h_lane_update(d, s, t, node, r, E[k].w3, (uint32_t)(k + 1) < cnt);
          if (st != H_UPD_DONE) {
            deferred = fpDefer = true;  // the item writes the event's result (the fast path's sends are the event's records)
            fpEvent = e;
            fpLevel = (uint32_t)H_ARG_LV(E[k].w3);
            if (st == H_UPD_BAIL) bailAt = k;
          }
        }
        if (!deferred) gst(d.evRes + e, res);
      }
      h_emit_job(s, job);  // a wide payload: a job of k_handel_copy
    }
    if (mine) {
      uint32_t WG_G* hdr = h_hdr(s, node);
      if (r.total != r.total0) hdr[HH_TOTAL] = (uint32_t)r.total;
      if (r.doneAt != r.doneAt0) d.nodes.doneAt[node] = r.doneAt;
      // (piece 12..15 is fetched only now — the line is in L2 since the visit's first load — so that it does not hold
      // four registers through the events: the kernel sits at the edge of three waves per SIMD)
      U4 h3;
      h3.x = h3.y = h3.z = h3.w = 0;
      if (nRecv || r.qdirty) h3 = gld((const U4 WG_G*)(hdr + 12));
      h_lane_store_header(hdr, r, h0, h2, h3, nRecv, bRecv);
    }
    {  // the node's wide updateVerifiedSignatures: an item of k_handel_update
      bool upd = false;
      uint32_t ue = 0, ua = 0, trail = 0;
#pragma unroll
      for (int k = 0; k < INBOX_SLOTS; k++)
        if (mine && !toDown && k == wideAt) {
          upd = true;
          ue = E[k].e;
          ua = E[k].w3;
        }
      {  // ... or the first event: the node's other events are k_handel_lane2's, listed here (one atomic per wavefront)
        const uint64_t tm = __ballot(updFirst);
        if (tm) {
          uint32_t tb = 0;
          const int leader = __ffsll((unsigned long long)tm) - 1;
          if (lane == leader) tb = atomicAdd(F(s.trailCount + 0), (uint32_t)__popcll(tm));
          tb = lane_bcast(tb, leader);
          if (updFirst) {
            const uint32_t idx = tb + __popcll(tm & lanes_lt());
            U4 q;
            q.x = (uint32_t)node;
            q.y = (vflags << 8) | cnt;
            q.z = E[0].e;
            q.w = 0u;
            gst(s.itemsTrail + idx, q);
            upd = true;
            ue = E[0].e;
            ua = E[0].w3;
            trail = (idx + 1u) << 16;
          }
        }
      }
      const uint64_t m = __ballot(upd);
      if (m) {
        uint32_t bb = 0;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if (lane == leader) bb = atomicAdd(F(s.updCount + 0), (uint32_t)__popcll(m));
        bb = lane_bcast(bb, leader);
        if (upd) {
          U4 q;
          q.x = (uint32_t)node;
          q.y = ue;
          q.z = ua;
          // (down | partition << 1 | (index of the node's itemsTrail entry + 1) << 9: 0 = nothing behind the update)
          q.w = (vflags & 1u) | ((vflags >> 8) << 1) | (trail >> 7);
          gst(s.itemsUpd + (bb + __popcll(m & lanes_lt())), q);
        }
      }
    }
    {  // deferred fast paths: items of the wave-per-node kernel, behind the node visits
      const uint64_t m = __ballot(fpDefer);
      if (m) {
        uint32_t bb = 0;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if (lane == leader) bb = atomicAdd(F(&d.g->nActiveB), (uint32_t)__popcll(m));
        bb = lane_bcast(bb, leader);
        if (fpDefer) {
          U4 q;
          q.x = (uint32_t)node;
          if (bailAt >= 0) {  // the visit's events from the bailAt-th (in event order) on
            q.y = HW_VISIT | (vflags << 8);
            q.z = cnt;
            q.w = (uint32_t)bailAt;
          } else {
            q.y = HW_FASTPATH | (vflags << 8);
            q.z = fpEvent;
            q.w = fpLevel;
          }
          gst(work + (bb + __popcll(m & lanes_lt())), q);
        }
      }
    }
    KPROF_MARK(d.g, 26);  // the lanes' events
  }
}

// updateVerifiedSignatures (:690-754) of a WIDE level (block of 8 .. 256+ words), one wavefront per task, lanes =
// 64-bit words of the level's block. The task touches its own level only, so it reads exactly that — two pieces of the
// node's header, the level's four scalars, its queue record, the signature and the level's three bitsets — not the
// node's whole record; its fast path, if one follows (:738-749), becomes an item of k_handel_wave, which runs next.
template <int WPE>
__global__ void __launch_bounds__(256, WPE) k_handel_update(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  const int lane = WG_LANE;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t nItems = *s.updCount;
  const int32_t t = d.g->now;
  if (wave >= nItems) return;
  U4 cur = gld(s.itemsUpd + wave);
  for (uint32_t q = wave; q < nItems; q += nWaves) {
    U4 nxt = cur;
    if (q + nWaves < nItems) nxt = gld(s.itemsUpd + (q + nWaves));
    const int32_t node = (int32_t)WG_READFIRST(cur.x);
    const uint32_t e = WG_READFIRST(cur.y), arg = WG_READFIRST(cur.z);
    const uint32_t wq = WG_READFIRST(cur.w);
    const uint32_t vflags = (wq & 1u) | (((wq >> 1) & 0xFFu) << 8), trailIdx = wq >> 9;  // (itemsTrail index + 1: deliveries wait behind this update)
    cur = nxt;
    const int pk = H_ARG_PK(arg), lv = H_ARG_LV(arg), slot = H_ARG_SLOT(arg);
    const int32_t from = H_ARG_FROM(arg);
    uint32_t WG_G* hdr = h_hdr(s, node);
    const Lv v = sib_view(node, lv);
    uint64_t WG_G* ti = h_row(s, node, HK_TI, lv);
    uint64_t WG_G* la = h_row(s, node, HK_LA, lv);
    uint64_t WG_G* vi = h_row(s, node, HK_VI, lv);
    const uint64_t WG_G* sig = h_sig_ptr(s, node, lv, slot);
    const int jF = (from >> 6) - v.bw;  // `from` lies in the level's block
    const uint64_t bit = 1ULL << (from & 63);
    uint64_t WG_G* qr = h_qrec(s, node, lv);
    uint64_t WG_G* ent = qr + H_QENT;
    // ---- every load of the task (addresses from its argument alone), before the first use. The kernel is bound by its
    // wave-level memory instructions: the six 16-byte pieces of the header / level / record are ONE instruction (lane k
    // fetches piece k), the rows and the signature move two words a lane.
    const U4 WG_G* piece = lane == 0 ? (const U4 WG_G*)(hdr + 4)             // .w: the sum of |totalIncoming| over the levels
                           : lane == 1 ? (const U4 WG_G*)(hdr + 8)           // .y .z: doneAt
                           : lane == 2 ? (const U4 WG_G*)(hdr + HH_PEND)
                           : lane == 3 ? (const U4 WG_G*)(hdr + HH_PENDFROM)
                           : lane == 4 ? (const U4 WG_G*)h_lv(s, node, HP_POS, lv) : (const U4 WG_G*)qr;
    U4 pg;
    pg.x = pg.y = pg.z = pg.w = 0;
    if (lane < 6) pg = gld(piece);
    // (the list: its first 13 entries are the record's first two lines; a longer one is fetched once its length is known)
    uint64_t entAll = lane < 13 ? ent[lane] : 0ULL;
    V2 sg[2], tiw[2], law[2], viw[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = 2 * (u * 64 + lane);
      sg[u].x = sg[u].y = tiw[u].x = tiw[u].y = law[u].x = law[u].y = viw[u].x = viw[u].y = 0;
      if (j < v.nw) {
        sg[u] = gld((const V2 WG_G*)(sig + j));
        law[u] = gld((const V2 WG_G*)(la + j));
        viw[u] = gld((const V2 WG_G*)(vi + j));
      }
      tiw[u].x = law[u].x | viw[u].x;  // totalIncoming as it is in memory = lastAggVerified | verifiedInd (the row is not read)
      tiw[u].y = law[u].y | viw[u].y;
    }
    auto bc = [&](uint32_t w, int l_) { return WG_READLANE(w, l_); };
    U4 hT, hD, pend, pfrom, a;
    hT.x = bc(pg.x, 0), hT.y = bc(pg.y, 0), hT.z = bc(pg.z, 0), hT.w = bc(pg.w, 0);
    hD.x = bc(pg.x, 1), hD.y = bc(pg.y, 1), hD.z = bc(pg.z, 1), hD.w = bc(pg.w, 1);
    pend.x = bc(pg.x, 2), pend.y = bc(pg.y, 2), pend.z = bc(pg.z, 2), pend.w = bc(pg.w, 2);
    pfrom.x = bc(pg.x, 3), pfrom.y = bc(pg.y, 3), pfrom.z = bc(pg.z, 3), pfrom.w = bc(pg.w, 3);
    a.x = bc(pg.x, 4), a.y = bc(pg.y, 4), a.z = bc(pg.z, 4), a.w = bc(pg.w, 4);
    HQHead qh;
    qh.len = (uint64_t)bc(pg.x, 5) | ((uint64_t)bc(pg.y, 5) << 32);
    qh.used = (uint64_t)bc(pg.z, 5) | ((uint64_t)bc(pg.w, 5) << 32);
    U4 WG_G* lvA = (U4 WG_G*)h_lv(s, node, HP_POS, lv);
    EvRes res;
    res.nrec = 0;
    res.ndraw = 0;
    const uint32_t pe = pk == 0 ? pend.x : pk == 1 ? pend.y : pk == 2 ? pend.z : pend.w;
    const uint32_t pf = pk == 0 ? pfrom.x : pk == 1 ? pfrom.y : pk == 2 ? pfrom.z : pfrom.w;
    const bool toDown = (vflags & VD_DOWN) != 0;
    if (toDown) {  // (:606 — a stopped node's task is consumed, not run; k_handel_lane2 consumes the deliveries behind it)
      if (lane == 0) gst(d.evRes + e, res);
      continue;
    }
    res.nrec = EV_TASK_RUN;
    if (pe != h_pend_word(lv, slot) || (int32_t)pf != from || v.nw > H_UPD_NW) {
      if (lane == 0) {
        set_err(d.g, ERR_PROTOCOL);
        gst(d.evRes + e, res);
      }
      continue;
    }
    // the VI / TI words holding `from`: with the lane (and half) that owns block word jF
    const int fLane = (jF >> 1) & 63, fU = jF >> 7, fHalf = jF & 1;
    uint64_t viF = 0, tiF = 0;
#pragma unroll
    for (int u = 0; u < 2; u++)
      if (fU == u) {
        viF = lane_bcast64(fHalf ? viw[u].y : viw[u].x, fLane);
        tiF = lane_bcast64(fHalf ? tiw[u].y : tiw[u].x, fLane);
      }
    const bool hadVI = (viF & bit) != 0, hadTI = (tiF & bit) != 0;
    const bool ownerLane = lane == fLane;
    int cVI = (int)a.w + (hadVI ? 0 : 1);
    const int cTI0 = (int)a.y;
    int cTI = cTI0, cLA = (int)a.z;
    bool improved = false;
    if (!hadTI) {
      cTI++;
      improved = true;
    }
    V2 viN[2];
    uint64_t acc = 0;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      viN[u] = viw[u];
      if (ownerLane && fU == u) {
        if (fHalf)
          viN[u].y |= bit;
        else
          viN[u].x |= bit;
      }
      acc += (uint64_t)(__popcll(sg[u].x | viN[u].x) + __popcll(sg[u].y | viN[u].y)) |
             ((uint64_t)(((sg[u].x & law[u].x) | (sg[u].y & law[u].y)) != 0) << 32);
    }
    acc = wave_sum64(acc);
    const int u2 = (int)(acc & 0xFFFFFFFFu);
    const bool inter = (acc >> 32) != 0;
    const bool replace = u2 > cVI;  // all.cardinality() > verifiedIndSignatures.cardinality()
    if (replace) {  // what the level's sets become — counted before anything is stored
      improved = true;
      uint64_t cnt = 0;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const uint64_t nlx = (inter ? 0ULL : law[u].x) | sg[u].x, nly = (inter ? 0ULL : law[u].y) | sg[u].y;
        if (2 * (u * 64 + lane) < v.nw)
          cnt += (uint64_t)(__popcll(nlx) + __popcll(nly)) | ((uint64_t)(__popcll(nlx | viN[u].x) + __popcll(nly | viN[u].y)) << 32);
      }
      cnt = wave_sum64(cnt);
      cLA = (int)(cnt & 0xFFFFFFFFu);
      cTI = (int)(cnt >> 32);
    }
    // A level that completes lets the levels above take the fast path (:738-749), whose scan of the emission lists reads
    // finishedPeers — which a delivery BEHIND this update may set. With deliveries behind it (trailCnt) such an update is not
    // this kernel's: nothing has been stored, the whole visit goes to k_handel_wave, which runs the fast path in its place.
    if (trailIdx && improved && cTI == v.size && s.p.fastPath > 0 && lv + 1 < s.L) {
      if (lane == 0) {
        U4 WG_G* tr = (U4 WG_G*)s.itemsTrail + (trailIdx - 1u);
        const uint32_t cntAll = gld(tr).y & 0xFFu;
        tr->w = 1u;  // (k_handel_lane2 leaves the node alone)
        const uint32_t bb = atomicAdd(F(&d.g->nActiveB), 1u);
        U4 qd;
        qd.x = (uint32_t)node;
        qd.y = HW_VISIT | (vflags << 8);
        qd.z = cntAll;
        qd.w = 0u;
        gst((U4 WG_G*)(VisitDesc WG_G*)d.activeB + bb, qd);
      }
      continue;
    }
    if (replace) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int j = 2 * (u * 64 + lane);
        if (j < v.nw) {  // (only the 16-byte pieces that change are written)
          V2 nla, nti;
          nla.x = (inter ? 0ULL : law[u].x) | sg[u].x;
          nla.y = (inter ? 0ULL : law[u].y) | sg[u].y;
          nti.x = nla.x | viN[u].x;
          nti.y = nla.y | viN[u].y;
          if (nla.x != law[u].x || nla.y != law[u].y) gst((V2 WG_G*)(la + j), nla);
          if (nti.x != tiw[u].x || nti.y != tiw[u].y) gst((V2 WG_G*)(ti + j), nti);
        }
      }
    } else if (!hadTI && ownerLane) {
      ti[jF] = tiF | bit;
    }
    if (!hadVI && ownerLane) vi[jF] = viF | bit;
    // toVerifyAgg.remove(vs): identity remove, sigQueueSize untouched
    const int len = (int)qh.len;
    if (len > 13) entAll = ent[lane];
    const uint64_t myEnt = lane < len ? entAll : ~0ULL;
    int newLen = len;
    bool emptied = false;
    {
      const uint64_t hit = __ballot(lane < len && (int)(myEnt & 0xFF) == slot);
      if (hit) {
        const int at = __ffsll((unsigned long long)hit) - 1;
        const uint64_t next = shfl64(myEnt, (lane + 1) & 63);
        if (lane >= at && lane < len - 1) ent[lane] = next;
        newLen = len - 1;
        emptied = len == 1;
      }
    }
    const int total = (int)hT.w + (cTI - cTI0);
    long long doneAt = (long long)((unsigned long long)hD.y | ((unsigned long long)hD.z << 32));
    const bool justDone = improved && doneAt == 0 && total >= s.p.threshold;
    const bool fastPathFollows = improved && cTI == v.size && s.p.fastPath > 0 && lv + 1 < s.L;
    if (lane == 0) {  // (toVerifyInd.set(from, false) :701: the set is SEEN & ~verifiedInd — nothing to store)
      hdr[HH_PEND + pk] = 0;
      a.y = (uint32_t)cTI;
      a.z = (uint32_t)cLA;
      a.w = (uint32_t)cVI;
      gst(lvA, a);
      if (total != (int)hT.w) hdr[HH_TOTAL] = (uint32_t)total;
      if (emptied) atomicAnd(F(hdr + HH_QMASK), ~(1u << lv));
      if (improved || !hadVI) qr[H_QVALID] = 0;  // one of the level's three sets changed: every cached evaluation of its entries is void
      if (newLen > 0) atomicOr(F(hdr + HH_QDIRTY), 1u << lv);  // (the list lost an entry: the level's summary is stale either way)
      // the entry's slot dies with this task unless another registered task still references it
      const uint32_t key = h_pend_word(lv, slot);
      const bool held = (pk != 0 && pend.x == key) || (pk != 1 && pend.y == key) || (pk != 2 && pend.z == key) || (pk != 3 && pend.w == key);
      HQHead nh;
      nh.len = (uint64_t)newLen;
      nh.used = held ? qh.used : (qh.used & ~(1ULL << slot));
      if (nh.len != qh.len || nh.used != qh.used) gst((HQHead WG_G*)qr, nh);
      if (justDone) {
        hdr[HH_DONE_LO] = (uint32_t)t;
        hdr[HH_DONE_HI] = 0;
        d.nodes.doneAt[node] = (long long)t;
      }
      if (fastPathFollows) {  // k_handel_wave runs it (and writes the event's result: its sends are the event's records)
        const uint32_t bb = atomicAdd(F(&d.g->nActiveB), 1u);
        U4 qd;
        qd.x = (uint32_t)node;
        qd.y = HW_FASTPATH | (vflags << 8);
        qd.z = e;
        qd.w = (uint32_t)lv;
        gst((U4 WG_G*)(VisitDesc WG_G*)d.activeB + bb, qd);
      } else {
        gst(d.evRes + e, res);
      }
    }
  }
}

// The SendSigs deliveries BEHIND a wide update that was the node's first event: one LANE per such node, after
// k_handel_update applied the update (it reads the header as that kernel left it: doneAt, the queue mask). The same
// h_lane_message as k_handel_lane; wide payloads become jobs of k_handel_copy, which runs next.
__global__ void __launch_bounds__(256) k_handel_lane2(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab, int behindDissem) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  const int lane = WG_LANE;
  // (behindDissem: the list of the nodes whose first event was their dissemination, launched behind k_handel_dissem)
  const U4 WG_G* items = behindDissem ? (const U4 WG_G*)s.itemsTrail2 : (const U4 WG_G*)s.itemsTrail;
  const uint32_t nItems = behindDissem ? *s.trail2Count : *s.trailCount;
  const int32_t t = d.g->now;
  const uint32_t stride = wgGx * blockDim.x;
  for (uint32_t a0 = (wgBx * blockDim.x + threadIdx.x) & ~63u; a0 < nItems; a0 += stride) {
    const uint32_t a = a0 + (uint32_t)lane;
    U4 it;
    it.x = it.y = it.z = 0;
    it.w = 1u;
    if (a < nItems) it = gld(items + a);
    const bool have = a < nItems && it.w == 0u;
    const int32_t node = (int32_t)it.x;
    const uint32_t cnt = have ? (it.y & 0xFFu) : 0u, vflags = it.y >> 8, eUpd = it.z;
    InboxEntry E[INBOX_SLOTS];
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) E[k].e = 0xFFFFFFFFu, E[k].w0 = 0, E[k].w2 = 0, E[k].w3 = 0;
    U4 h0, h2;
    h0.x = h0.y = h0.z = h0.w = 0;
    h2 = h0;
    if (have) {
#pragma unroll
      for (int k = 0; k < INBOX_SLOTS; k++) E[k] = gld(d.inbox + ((size_t)node * INBOX_SLOTS + k));
      const uint32_t WG_G* hdr = h_hdr(s, node);
      h0 = gld((const U4 WG_G*)hdr);
      h2 = gld((const U4 WG_G*)(hdr + 8));
    }
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++)
      if ((uint32_t)k >= cnt || E[k].e == eUpd) E[k].e = 0xFFFFFFFFu;  // (the update itself: k_handel_update's)
#define H_CSWAP(A, B)                  \
  if (E[B].e < E[A].e) {               \
    const InboxEntry x = E[A];         \
    E[A] = E[B];                       \
    E[B] = x;                          \
  }
    H_CSWAP(0, 1) H_CSWAP(2, 3) H_CSWAP(0, 2) H_CSWAP(1, 3) H_CSWAP(1, 2)
#undef H_CSWAP
    HLaneNode r;
    r.doneAt = r.doneAt0 = (long long)((unsigned long long)h2.y | ((unsigned long long)h2.z << 32));
    r.startAt = (int32_t)h0.w;
    r.sigQueueSize = r.sigQueueSize0 = (int32_t)h0.y;
    r.msgFiltered = r.msgFiltered0 = (int32_t)h0.z;
    r.qmask = r.qmask0 = h2.w;
    r.qdirty = 0;
    r.total = r.total0 = 0;
    const bool toDown = (vflags & VD_DOWN) != 0;
    const uint8_t toPart = (uint8_t)(vflags >> 8);
    long long nRecv = 0, bRecv = 0;
#pragma unroll
    for (int k = 0; k < INBOX_SLOTS; k++) {
      CopyJob job;
      job.nw = 0;
      if (have && E[k].e != 0xFFFFFFFFu) {
        const int32_t from = (int32_t)(E[k].w0 & 0x0FFFFFFFu);
        EvRes res;
        res.nrec = 0;
        res.ndraw = 0;
        if (!toDown && (d.nparts == 0 || d.nodes.part[from] == toPart)) {  // C/Network.java:606
          nRecv++;
          bRecv += h_msg_size((int)(E[k].w2 & 31u));
          res.nrec = EV_DELIVERED | ((E[k].w2 & 31u) << 24);
          h_lane_message(d, s, t, node, r, from, E[k].w2, E[k].w3, job);
        }
        gst(d.evRes + E[k].e, res);
      }
      h_emit_job(s, job);
    }
    // (the header as k_handel_update / k_handel_dissem left it — this lane is the node's only writer in this launch)
    if (have) {
      U4 h3;
      h3.x = h3.y = h3.z = h3.w = 0;
      if (nRecv || r.qdirty) h3 = gld((const U4 WG_G*)(h_hdr(s, node) + 12));
      h_lane_store_header(h_hdr(s, node), r, h0, h2, h3, nRecv, bRecv);
    }
  }
}

// ... the wide SendSigs payloads k_handel_lane delivered (:769 `sig` of the SigToVerify: the receiver's level block of
// the sender's snapshot): one WAVEFRONT per job, lanes = consecutive 64-bit words, the next job's descriptor in flight
// while the current one is copied. (As the tail of the lane kernel — each wavefront copying its own lanes' payloads
// one round after the other — the copies were that kernel's longest dependent chain.)
__global__ void __launch_bounds__(256) k_handel_copy(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  const int lane = WG_LANE;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  // the small payloads (<= H_JOB_SMALL words: levels of up to 4096 ids — most of the jobs): EIGHT LANES per job, two words a
  // lane and round, eight jobs per wavefront. (One wavefront per job left most of its lanes idle for them, and the kernel's
  // duration is jobs / resident wavefronts x a round trip.)
  const uint32_t nSmall = *s.jobSmallCount;
  for (uint32_t qb = wave * 8; qb < nSmall; qb += nWaves * 8) {
    const uint32_t q = qb + (uint32_t)(lane >> 3);
    if (q < nSmall) {
      const CopyJob job = gld(s.jobsSmall + q);
      for (int j = 2 * (lane & 7); j < job.nw; j += 16) {  // (nw is a power of two >= 2)
        const uint64_t a = job.src[j], b = job.src[j + 1];
        job.dst[j] = a;
        job.dst[j + 1] = b;
      }
    }
  }
  const uint32_t nJobs = *s.jobCount;
  if (wave >= nJobs) return;
  CopyJob cur = gld(s.jobs + wave);
  for (uint32_t q = wave; q < nJobs; q += nWaves) {
    CopyJob nxt = cur;
    if (q + nWaves < nJobs) nxt = gld(s.jobs + (q + nWaves));
    for (int j0 = 0; j0 < cur.nw; j0 += 256) {
      uint64_t v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int j = j0 + u * 64 + lane;
        v[u] = j < cur.nw ? cur.src[j] : 0ULL;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int j = j0 + u * 64 + lane;
        if (j < cur.nw) cur.dst[j] = v[u];
      }
    }
    cur = nxt;
  }
}

// The delivery pass, second kernel: one WAVEFRONT per work item of the list k_handel_lane wrote — a node visit
// (receiveUntil's body for the node's events in event order, from its inbox line) or a deferred fast path.
// Software-pipelined: while item a runs, the header and the inbox line of item a + nWaves and the descriptor of item
// a + 2 nWaves are in flight. A visit writes no other node's header (effects on other nodes travel as envelopes, at
// least one ms later), so fetching the next header early reads what the visit itself would.
// The delivery pass, dissemination tier: one wavefront per node whose first event of the ms is its dissemination task
// (P/Handel.java:331-343) — in a run with a synchronised start that is every live node once per period, 3/4 of a ms that
// costs three ordinary ones. Only that event: a kernel that holds nothing of updateVerifiedSignatures / onNewSig fits twice
// the wavefronts of k_handel_wave. The node's later events of the ms (SendSigs that were sent before the task was
// re-armed) become a visit of k_handel_wave that skips the first event.
template <int WPE>
__global__ void __launch_bounds__(256, WPE) k_handel_dissem(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  __shared__ LevelScalars shP[4];
  const int lane = WG_LANE, w = threadIdx.x >> 6;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t nItems = *s.disCount;
  const int32_t t = d.g->now;
  for (uint32_t a = wave; a < nItems; a += nWaves) {
    const U4 it = gld((const U4 WG_G*)s.itemsDis + a);
    const int32_t node = (int32_t)WG_READFIRST(it.x);
    const uint32_t vflags = WG_READFIRST(it.y) >> 8, cnt = WG_READFIRST(it.y) & 0xFFu;
    const uint32_t e = WG_READFIRST(it.z), w0 = WG_READFIRST(it.w);
    const HandelProto::Pre hdr = HandelProto::prefetch(s, node);
    Ctx c{d, t, node, 0, 0, 0, 0, 0, 0, 0};
    HandelProto::NodeRegs r;
    HandelProto::node_begin_pre(c, s, r, &shP[w], hdr);
    c.ev = e;
    c.sub = 0;
    c.draws = 0;
    c.outBase = w0 & 0x0FFFFFFFu;
    c.outCap = d.boundTask[0] + 1u;
    uint32_t flags = 0;
    if (!(vflags & VD_DOWN)) {  // (C/Network.java:606: a task's `from` is its own node)
      flags = EV_TASK_RUN;
      HandelProto::dissemination(c, s, r);
      // PeriodicTask.action re-arm (C/messages/PeriodicTask.java:39-47)
      c.put(O_PERIODIC, node, H_TASK_DISSEMINATION, (uint32_t)s.p.disseminationPeriodMs, t + s.p.disseminationPeriodMs, 0, false);
    }
    if (lane == 0) {
      EvRes res;
      res.nrec = c.sub | flags | c.evFlags;
      res.ndraw = c.draws;
      gst(d.evRes + e, res);
    }
    HandelProto::node_counters(c, s, r, 0, 0);
    HandelProto::node_end(c, s, r);
    // (the rest of the node's events: k_handel_lane listed them as a visit of k_handel_wave that skips the first event)
    __builtin_amdgcn_wave_barrier();
  }
}

template <int WPE, bool ATK>
__global__ void __launch_bounds__(256, WPE) k_handel_wave(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab,
                                                          int disSkipped) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  typedef HandelProtoT<ATK> HP;
  __shared__ LevelScalars shP[4];
  const int lane = WG_LANE, w = threadIdx.x >> 6;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  // (the host did not launch k_handel_dissem for this ms — no member's dissemination task has this phase: its list must be empty)
  if (disSkipped && wave == 0 && lane == 0 && *s.disCount != 0) set_err(d.g, ERR_PROTOCOL);
  const uint32_t nWork = d.g->nActiveB;
  const int32_t t = d.g->now;
  const U4 WG_G* work = (const U4 WG_G*)(const VisitDesc WG_G*)d.activeB;
  uint32_t a = wave;
  if (a >= nWork) return;
  auto inbox_of = [&](const U4& desc) -> InboxEntry {  // lane k < 4: entry k of the node's line
    return gld(d.inbox + ((size_t)WG_READFIRST(desc.x) * INBOX_SLOTS + (lane < INBOX_SLOTS ? lane : 0)));
  };
  U4 descCur = gld(work + a);
  typename HP::Pre hdrCur = HP::prefetch(s, (int32_t)WG_READFIRST(descCur.x));
  InboxEntry inCur = inbox_of(descCur);
  U4 descNext = descCur;
  if (a + nWaves < nWork) descNext = gld(work + (a + nWaves));
  for (;;) {
    KPROF_DECL;
    KPROF_COUNT(d.g, 0);
    const uint32_t an = a + nWaves;
    const bool haveNext = an < nWork;
    typename HP::Pre hdrNext = hdrCur;
    InboxEntry inNext = inCur;
    U4 descNext2 = descNext;
    if (haveNext) {
      hdrNext = HP::prefetch(s, (int32_t)WG_READFIRST(descNext.x));
      inNext = inbox_of(descNext);
      if (an + nWaves < nWork) descNext2 = gld(work + (an + nWaves));
    }
    {
      const int32_t node = (int32_t)WG_READFIRST(descCur.x);
      const uint32_t kf = WG_READFIRST(descCur.y);
      Ctx c{d, t, node, 0, 0, 0, 0, 0, 0, 0};
      typename HP::NodeRegs r;
      HP::node_begin_pre(c, s, r, &shP[w], hdrCur);
      KPROF_MARK(d.g, 31);  // the next item's prefetches issued, this item's descriptor + header arrived
      if ((kf & 0xFFu) == HW_VISIT) {
        deliver_visit_inbox<HP>(d, s, c, r, node, WG_READFIRST(descCur.z), kf >> 8, inCur, WG_READFIRST(descCur.w));
      } else {
        // the fast path of an updateVerifiedSignatures a lane applied (h_lane_update): its sends are that event's records
        const uint32_t e = WG_READFIRST(descCur.z);
        const EvAux aux = gld(d.evAux + e);
        c.ev = e;
        c.outBase = aux.outBase;
        c.outCap = aux.outCap;
        HP::fast_path(c, s, r.ls, (int)WG_READFIRST(descCur.w));
        if (lane == 0) {
          EvRes res;
          res.nrec = c.sub | EV_TASK_RUN;
          res.ndraw = c.draws;
          gst(d.evRes + e, res);
        }
        HP::node_counters(c, s, r, 0, 0);
        HP::store_levels(s, node, r.ls);  // posInLevel / outgoingFinished of the levels it sent for, the counters
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (!haveNext) break;
    a = an;
    descCur = descNext;
    hdrCur = hdrNext;
    inCur = inNext;
    descNext = descNext2;
  }
}

// the evaluation of ONE queue entry against the level's state — what bestToVerify (:570-634) asks of it, and what stays
// true until the level's sets change (HandelState::qcache): bit 0 keep = sizeIfIncluded (:532-540) > |totalIncoming|,
// bits 1.. its score (:655-668), which counts if the entry's rank is inside the window
__device__ __forceinline__ uint32_t h_eval_word(int u1, int u2, int cs, bool iTI, bool iLA, int curSize, int cLA, int size) {
  const int sII = iTI ? u2 : u1;  // sizeIfIncluded :532-540
  const int score = cLA >= size ? 0 : (!iLA ? cLA + cs : max(0, u2 - cLA));  // score(l, sig) :655-668
  return (sII > curSize ? 1u : 0u) | ((uint32_t)score << 1);
}
// rank <= windowIndex + currWindowSize (:597) in Java's int arithmetic (a saturated rank makes the sum wrap)
__device__ __forceinline__ bool h_in_window(int rank, int windowIndex, int window) {
  return rank <= (int)((uint32_t)windowIndex + (uint32_t)window);
}

// checkSigs looks at every level with a queue (:796-806), but only ONE level's candidate is used — the one the draw among
// the levels that HAVE a candidate picks (:788-790). Whether a level has one follows from three numbers of its list while
// the list and the level's sets stay as k_handel_a1 last saw them (the level is "clean", HH_QDIRTY): every listed entry
// is then one the last curation kept, so windowIndex = the smallest rank, and bestToVerify returns something iff some
// entry has a positive score (inside the window it is bestInside's, outside it bestOutside exists) or some entry lies
// outside the window (largest rank > smallest rank + currWindowSize). The summary: HP_SPARE0 = smallest rank | (some
// score > 0) << 31, HP_SPARE1 = largest rank (the two planes byzantineSuicide uses are free without the attack).
// k_handel_cond_pre answers "has a candidate" for the clean levels from it; which entry it is is computed only for the
// level the draw picked (h_pick_cached, k_handel_cond_a2).
__device__ __forceinline__ bool h_summary_has_candidate(uint32_t sum0, uint32_t sum1, int window) {
  return (sum0 >> 31) || !h_in_window((int)sum1, (int)(sum0 & 0x7FFFFFFFu), window);
}
// bestToVerify (:570-634) of a CLEAN level by one lane: every entry's evaluation is cached and kept. Returns signer << 8 | slot, or -1
__device__ __forceinline__ long long h_pick_cached(const HandelState& s, int32_t node, int l, int window) {
  const uint64_t WG_G* qr = h_qrec(s, node, l);
  const uint64_t WG_G* ent = qr + H_QENT;
  const uint32_t WG_G* cache = s.qcache + ((size_t)node * s.L + l) * (size_t)s.QC;
  const HQHead qh = gld((const HQHead WG_G*)qr);
  const uint64_t valid = qr[H_QVALID];
  uint64_t e4[4];
#pragma unroll
  for (int i = 0; i < 4; i++) e4[i] = ent[i];
  const U4 c4 = gld((const U4 WG_G*)cache);
  const int len = (int)qh.len;
  int windowIndex = INT32_MAX;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (i < len) windowIndex = min(windowIndex, (int)(uint32_t)(e4[i] >> 32));
  for (int i = 4; i < len; i++) windowIndex = min(windowIndex, (int)(uint32_t)(ent[i] >> 32));
  long long bestInside = -1, bestOutside = -1;
  int bestScore = 0, bestOutsideRank = 0;
  bool ok = true;
  auto consider = [&](uint64_t x) {
    const int slot = (int)(x & 0xFF);
    const uint32_t cw = slot == 0 ? c4.x : slot == 1 ? c4.y : slot == 2 ? c4.z : slot == 3 ? c4.w : cache[slot];
    ok = ok && ((valid >> slot) & 1ULL) && (cw & 1u);
    const int rank = (int)(uint32_t)(x >> 32);
    if (h_in_window(rank, windowIndex, window)) {
      const int score = (int)(cw >> 1);
      if (score > bestScore) {
        bestScore = score;
        bestInside = (long long)(uint32_t)x;
      }
    } else if (bestOutside < 0 || rank < bestOutsideRank) {
      bestOutside = (long long)(uint32_t)x;
      bestOutsideRank = rank;
    }
  };
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (i < len) consider(e4[i]);
  for (int i = 4; i < len; i++) consider(ent[i]);
  if (!ok) return -1;  // (not a clean level: the caller reports it)
  return bestInside >= 0 ? bestInside : bestOutside;
}

// ---- conditional-task phase (C/Network.java:543-566 driving HNode.checkSigs :796-837) -------------
// PRE: which conditional tasks run at this edge — one lane per node, coalesced reads of the two dense words that
// decide it. checkSigs looks at every level with a queue, and bestToVerify of one level (:570-634) depends on nothing
// of the others: every (runner, level with a non-empty queue) becomes an ITEM of k_handel_a1, in one of two lists by
// the lanes its block needs.
__global__ void __launch_bounds__(256) k_handel_cond_pre(const EngineDev* __restrict__ tab,
                                                         const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  const int32_t t = d.g->now, until = d.g->until;
  const uint32_t epoch = d.g->epoch;
  const uint32_t stride = wgGx * blockDim.x;
  const int lane = WG_LANE;
  __shared__ uint32_t shTot[2][4], shBase[2];
  if (wgBx == 0 && threadIdx.x == 0) {  // (the delivery pass's copy jobs and wide updates have been done)
    *s.jobCount = 0;
    *s.jobSmallCount = 0;
    *s.updCount = 0;
    *s.disCount = 0;
    *s.trailCount = 0;
    *s.trail2Count = 0;
  }
  for (uint32_t n0 = (uint32_t)s.lo + wgBx * blockDim.x; n0 < (uint32_t)s.hi; n0 += stride) {
    const uint32_t node = n0 + threadIdx.x;
    // nextMessage(): drop from the private copy if minStartTime > until or the node is down; evaluate
    // at most once per call (epoch); evaluate only when minStartTime <= time.
    bool run = false;
    uint32_t qm = 0, qd = 0;
    if (node < (uint32_t)s.hi) {
      uint32_t WG_G* ct = s.ct + 2 * (size_t)node;
      uint32_t WG_G* h = h_hdr(s, (int32_t)node);
      const uint32_t ctMin = ct[0], ctEpoch = ct[1];
      if (!d.nodes.down[node] && ctEpoch != epoch) {
        const int32_t ms = (int32_t)ctMin;
        if (ms <= until && ms <= t) {
          ct[1] = epoch;
          run = h[HH_SIGQ] != 0;  // startIf = hasSigToVerify (:345-347)
        }
      }
      if (run) {
        ct[0] = (uint32_t)(t + (int32_t)h[HH_PAIR]);  // minStartTime = time + duration (:557-560)
        // sigQueueSize drifts above the real queue lengths: checkSigs then runs over empty
        // lists, finds no candidate, draws nothing and changes nothing (:800-806) — such a node has no item
        qm = h[HH_QMASK] & ~1u;
        // levels holding an entry whose evaluation is not cached: this edge's k_handel_a1 evaluates them all (HandelState::qcache)
        qd = h[HH_QDIRTY];
        if (qd) h[HH_QDIRTY] = 0;
      }
      // the clean levels: whether bestToVerify would return something, from the level's summary — no item, no list read.
      // Which entry it is is computed for the level the draw picks (k_handel_cond_a2: h_pick_cached)
      uint32_t cmClean = 0;
      if (s.atk) qd = qm;  // (an attack's run: every level an item, nothing cached)
      if (qm & ~qd) {
        const int window = (int)h[HH_WINDOW];
        for (uint32_t m = qm & ~qd; m; m &= m - 1) {
          const int l = __ffs(m) - 1;
          const U4 b = gld((const U4 WG_G*)h_lv(s, (int32_t)node, HP_CAND, l));  // {candidate, outgoingFinished, summary}
          if (h_summary_has_candidate(b.z, b.w, window)) cmClean |= 1u << l;
        }
      }
      s.candMask[node] = cmClean;
      s.cleanMask[node] = cmClean;
      qm &= qd;  // the items: the levels with something to evaluate
    }
    // the node's items: one per level with a queue, appended BLOCK-aggregated — one atomic per list and block. (One per
    // wavefront was 512 same-address atomics per list and engine in every ms: an L2 atomic unit retires ~ 88 of those per
    // us, and every wavefront waited for its turn: 22 -> 15 us per ordinary ms at 24 copies.)
    // A level with something to evaluate is a lane's item while its block is <= H_LANE_NW words and a wavefront's beyond.
    // A wavefront's level whose verified sets are still empty (|lastAggVerified| = |verifiedInd| = 0, hence totalIncoming
    // too) says so in its item (bit 31): the evaluation then needs the signatures alone, not the level's rows.
    uint32_t mLane = 0, mWave = 0, mEmpty = 0;
    for (uint32_t m = qm; m; m &= m - 1) {
      const int l = __ffs(m) - 1;
      if (!s.atk && h_nw(l) <= s.laneNw) {  // (an attack's run: every item by a wavefront)
        mLane |= 1u << l;
      } else {
        mWave |= 1u << l;
        if (!s.atk) {
          const U4 a = gld((const U4 WG_G*)h_lv(s, (int32_t)node, HP_POS, l));
          if ((a.z | a.w) == 0) mEmpty |= 1u << l;
        }
      }
    }
    const int w = (int)(threadIdx.x >> 6);
    uint32_t incl2[2], mine2[2];
#pragma unroll
    for (int which = 0; which < 2; which++) {
      mine2[which] = (uint32_t)__popc(which == 0 ? mLane : mWave);
      incl2[which] = wave_incl_scan32(mine2[which]);
      if (lane == 63) shTot[which][w] = incl2[which];
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      const int which = (int)threadIdx.x;
      const uint32_t tot = shTot[which][0] + shTot[which][1] + shTot[which][2] + shTot[which][3];
      shBase[which] = tot ? atomicAdd(F(s.itemCount + which), tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 2; which++) {
      const uint32_t mm = which == 0 ? mLane : mWave;
      uint32_t base = shBase[which] + incl2[which] - mine2[which];
      for (int k = 0; k < 4; k++)
        if (k < w) base += shTot[which][k];
      uint32_t WG_G* list = which == 0 ? (uint32_t WG_G*)s.itemsLane : (uint32_t WG_G*)s.itemsWave;
      for (uint32_t m = mm; m; m &= m - 1) {
        const int l = __ffs(m) - 1;
        list[base++] = node | ((uint32_t)l << 24) | (which == 1 && ((mEmpty >> l) & 1u) ? 0x80000000u : 0u);
      }
    }
    __syncthreads();  // (shTot / shBase are rewritten by the next round)
  }
}

// the end of an item: the curated list's bookkeeping and the level's candidate (`cand`: signer << 8 | slot of the chosen
// entry, or -1). `relMask`: queue slots of dropped entries no registered task holds; several items of one node run
// concurrently (other levels), so the node's shared words are updated with atomics; the record's head is the item's own.
// `sum0 / sum1`: the level's summary over the entries that stay listed (h_summary) — what the following edges read
// instead of the list while the level stays clean.
__device__ __forceinline__ void h_item_finish(const HandelState& s, int32_t node, int l, uint64_t WG_G* qr, HQHead qh, int len,
                                              int kept, unsigned long long relMask, long long cand, uint32_t sum0, uint32_t sum1) {
  uint32_t WG_G* hdr = h_hdr(s, node);
  if (!s.atk && kept > 0) {
    *h_lv(s, node, HP_SPARE0, l) = sum0;
    *h_lv(s, node, HP_SPARE1, l) = sum1;
  }
  if (kept != len) {  // replaceToVerifyAgg :636-646
    qh.len = (uint64_t)kept;
    qh.used &= ~relMask;
    gst((HQHead WG_G*)qr, qh);
    atomicAdd(F(hdr + HH_SIGQ), (uint32_t)(kept - len));  // sigQueueSize -= dropped
    if (kept == 0) atomicAnd(F(hdr + HH_QMASK), ~(1u << l));
  }
  if (cand >= 0) {
    *h_lv(s, node, HP_CAND, l) = (uint32_t)cand;
    atomicOr(F(s.candMask + node), 1u << l);
  }
}

// bestToVerify (:570-634) of (node, l) by one wavefront: lanes = 64-bit words of the level's block while an entry is
// evaluated, lane i = entry i of the list for everything else. Curates the list and records the level's candidate
// (h_item_finish); returns it (signer << 8 | slot) or -1. Only the entries without a cached evaluation have their
// signature read (HandelState::qcache; an attack's run caches nothing and evaluates them all).
template <bool ATK>
__device__ __forceinline__ long long h_best_wave(const EngineDev& d, const HandelState& s, int32_t node, int l,
                                                 bool emptySets = false) {
  const int lane = WG_LANE;
  KPROF_DECL;
  KPROF_COUNT(d.g, 16);
  const uint32_t WG_G* hdr = h_hdr(s, node);
  const Lv v = sib_view(node, l);
  const uint64_t WG_G* ti = h_row(s, node, HK_TI, l);  // the level's block: word j of it is [j]
  const uint64_t WG_G* la = h_row(s, node, HK_LA, l);
  const uint64_t WG_G* vi = h_row(s, node, HK_VI, l);
  uint64_t WG_G* qr = h_qrec(s, node, l);
  uint64_t WG_G* ent = qr + H_QENT;
  uint32_t WG_G* cache = s.qcache + ((size_t)node * s.L + l) * (size_t)s.QC;
  // ---- everything the item's address alone decides, before the first use. These kernels are bound by the number of
  // wave-level memory instructions, so: the five 16-byte pieces (pending table, the header words with the window, the
  // level's scalars, the record's head, its valid mask) are ONE instruction — lane k fetches piece k —, the list is one
  // (lane i = entry i; the record's first two lines, a longer list is fetched once its length is known), and the rows and
  // signatures move two words a lane (blocks of <= 256 words: two instructions an array at most)
  const U4 WG_G* piece = lane == 0 ? (const U4 WG_G*)(hdr + HH_PEND)
                         : lane == 1 ? (const U4 WG_G*)(hdr + 4)
                         : lane == 2 ? (const U4 WG_G*)h_lv(s, node, HP_POS, l)
                         : lane == 3 ? (const U4 WG_G*)qr : (const U4 WG_G*)(qr + H_QVALID);
  U4 pg;
  pg.x = pg.y = pg.z = pg.w = 0;
  if (lane < 5) pg = gld(piece);
  uint64_t entAll = lane < 13 ? ent[lane] : 0ULL;
  const bool wideRound = v.nw <= 256;  // (levels <= 15; beyond: the word loop below)
  V2 ti2[2], vi2[2], la2[2];  // (ti2 = la2 | vi2: totalIncoming is their union, the row is not read)
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int j = 2 * (u * 64 + lane);
    vi2[u].x = vi2[u].y = la2[u].x = la2[u].y = 0;
    if (ATK && v.nw == 1) {  // (the levels below 8, which only the attack's runs bring here: one masked word)
      if (j == 0) {
        vi2[u].x = vi[0] & v.mask;
        la2[u].x = la[0] & v.mask;
      }
    } else if (wideRound && j < v.nw && !emptySets) {  // (emptySets: both rows are zero — cond_pre read their counts)
      vi2[u] = gld((const V2 WG_G*)(vi + j));
      la2[u] = gld((const V2 WG_G*)(la + j));
    }
    ti2[u].x = la2[u].x | vi2[u].x;
    ti2[u].y = la2[u].y | vi2[u].y;
  }
  U4 pend;
  pend.x = WG_READLANE(pg.x, 0);
  pend.y = WG_READLANE(pg.y, 0);
  pend.z = WG_READLANE(pg.z, 0);
  pend.w = WG_READLANE(pg.w, 0);
  const int window = (int)WG_READLANE(pg.y, 1);  // HH_WINDOW = word 5
  const int curSize = (int)WG_READLANE(pg.y, 2), cLA = (int)WG_READLANE(pg.z, 2);
  HQHead qh;
  qh.len = (uint64_t)WG_READLANE(pg.x, 3) | ((uint64_t)WG_READLANE(pg.y, 3) << 32);
  qh.used = (uint64_t)WG_READLANE(pg.z, 3) | ((uint64_t)WG_READLANE(pg.w, 3) << 32);
  const uint64_t valid0 = ATK ? 0ULL : ((uint64_t)WG_READLANE(pg.x, 4) | ((uint64_t)WG_READLANE(pg.y, 4) << 32));
  const int len = (int)qh.len;
  if (len > 13) entAll = ent[lane];
  const uint64_t myEnt = lane < len ? entAll : ~0ULL;
  const int mySlot = lane < len ? (int)(myEnt & 0xFF) : 0;
  const int myRank = lane < len ? (int)(uint32_t)(myEnt >> 32) : INT32_MAX;
  const bool myValid = lane < len && ((valid0 >> mySlot) & 1ULL);
  uint32_t myCw = myValid ? cache[mySlot] : 0u;  // keep | score << 1 of my entry (the others: evaluated below)
  KPROF_MARK(d.g, 19);  // the item's header pieces, list and row words
  const int windowIndex = wave_reduce_min_i32(myRank);  // Collections.min(rank)
  uint64_t blkM = 0;  // entries whose signer is blacklisted (:592 `!blacklist.get(stv.from)`)
  if (ATK) {
    // ---- createSuicideByzantineSig (:538-559, called by bestToVerify :577-584): the first byzantine (down), not yet
    // blacklisted peer of the level from suicideBizAfter on whose rank is inside the window sends a bad signature over
    // the whole block. The scan is the reference's loop, 64 peers a step.
    const int sba = (int)*h_lv(s, node, HP_SPARE0, l);
    if (sba >= 0 && len > 0) {
      const int size = 1 << (l - 1);
      const size_t peers0 = (size_t)node * (s.N - 1) + (size - 1);
      const int maxRank = windowIndex + window;
      int firstCond = -1, hit = -1, hitRank = 0;
      for (int i0 = sba; i0 < size && hit < 0; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < size;
        const int32_t p = in ? h_peer(s, peers0 + i) : 0;
        const bool cond = in && d.nodes.down[p] && !HandelProtoT<true>::blk(s, node, p);
        const int32_t rk = cond ? s.ranks[(size_t)node * s.N + p] : 0;
        const uint64_t cm = __ballot(cond);
        if (firstCond < 0 && cm) firstCond = i0 + __ffsll((unsigned long long)cm) - 1;
        const uint64_t hm = __ballot(cond && rk < maxRank);
        if (hm) {
          const int src = __ffsll((unsigned long long)hm) - 1;
          hit = (int)lane_bcast((uint32_t)p, src);
          hitRank = (int)lane_bcast((uint32_t)rk, src);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) *h_lv(s, node, HP_SPARE0, l) = (uint32_t)firstCond;  // (-1: no byzantine nodes left in this level)
      if (hit >= 0) {  // toVerifyAgg.add(bSig); sigQueueSize++; return bSig — no curation this time
        const int qc = h_qcap(s, l);
        const unsigned long long capMask = qc >= 64 ? ~0ULL : ((1ULL << qc) - 1ULL);
        const unsigned long long freeM = ~qh.used & capMask;
        if (freeM == 0 || len >= 64) {
          if (lane == 0) set_err(d.g, ERR_QUEUE_CAP);
          return -1;
        }
        const int slot = __ffsll(freeM) - 1;
        uint64_t WG_G* dst = h_sig_ptr(s, node, l, slot);
        H_FOR_WORDS(v, j) dst[j] = v.mask;  // sig = waitedSigs: the whole sibling block
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
          ent[len] = h_entry(hitRank, hit, slot);
          HQHead nh;
          nh.len = (uint64_t)(len + 1);
          nh.used = qh.used | (1ULL << slot);
          gst((HQHead WG_G*)qr, nh);
          qr[s.qBad] |= 1ULL << slot;
          atomicAdd(F((uint32_t WG_G*)hdr + HH_SIGQ), 1u);
          *h_lv(s, node, HP_CAND, l) = ((uint32_t)hit << 8) | (uint32_t)slot;
          atomicOr(F(s.candMask + node), 1u << l);
        }
        return ((long long)hit << 8) | (long long)slot;
      }
    }
    const bool myBlk = lane < len && HandelProtoT<true>::blk(s, node, (int32_t)((myEnt >> 8) & 0xFFFFFFu));
    blkM = __ballot(myBlk);
  }
  // ---- the entries without a cached evaluation: their signatures against the level's three sets, one after the other
  // (an item usually has ONE: the entry that arrived since the last edge — the kernel's residency is worth more than a
  // second signature in flight)
  const uint64_t inv = __ballot(lane < len && !myValid);
  uint64_t newValid = valid0;
  for (uint64_t m = inv; m; m &= m - 1) {
    const int pos = __ffsll((unsigned long long)m) - 1;
    const int slot = (int)(lane_bcast((uint32_t)myEnt, pos) & 0xFFu);
    const uint64_t WG_G* sig = h_sig_ptr(s, node, l, slot);
    uint64_t a = 0, b = 0;
    auto word = [&](uint64_t sgw, uint64_t tiw, uint64_t viw, uint64_t law) {
      a += (uint64_t)__popcll(sgw | tiw | viw) | ((uint64_t)__popcll(sgw | viw) << 21) | ((uint64_t)__popcll(sgw) << 42);
      b += (uint64_t)((sgw & tiw) != 0) | ((uint64_t)((sgw & law) != 0) << 21);
    };
    if (ATK && v.nw == 1) {
      if (lane == 0) word(sig[0], ti2[0].x, vi2[0].x, la2[0].x);
    } else if (wideRound) {
      V2 sg[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int j = 2 * (u * 64 + lane);
        sg[u].x = sg[u].y = 0;
        if (j < v.nw) sg[u] = gld((const V2 WG_G*)(sig + j));
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {  // (words beyond the block are zero in every array: they add nothing)
        word(sg[u].x, ti2[u].x, vi2[u].x, la2[u].x);
        word(sg[u].y, ti2[u].y, vi2[u].y, la2[u].y);
      }
    } else {
      H_FOR_WORDS(v, j) {
        const uint64_t viw = vi[j], law = la[j];
        word(sig[j], law | viw, viw, law);
      }
    }
    a = wave_sum64(a);
    b = wave_sum64(b);
    const int u1 = (int)(a & 0x1FFFFF), u2 = (int)((a >> 21) & 0x1FFFFF), cs = (int)((a >> 42) & 0x1FFFFF);
    const bool iTI = (b & 0x1FFFFF) != 0, iLA = ((b >> 21) & 0x1FFFFF) != 0;
    const uint32_t cw = h_eval_word(u1, u2, cs, iTI, iLA, curSize, cLA, v.size);
    if (lane == pos) {
      myCw = cw;
      if (!ATK) cache[slot] = cw;
    }
    newValid |= 1ULL << slot;
  }
  KPROF_ADD(d.g, 21, __popcll(inv));
  KPROF_ADD(d.g, 20, len);
  KPROF_MARK(d.g, 22);  // the entries (kprof21: how many were evaluated)
  // ---- the choice, lane i = entry i: the curated list keeps the entries that can improve the aggregate (:592); the best
  // inside the window is the FIRST entry with the strictly greatest positive score, the best outside the FIRST with the
  // smallest rank (:597-611)
  const bool myKeep = lane < len && (myCw & 1u) && !(ATK && ((blkM >> lane) & 1ULL));
  const bool myInside = myKeep && h_in_window(myRank, windowIndex, window);
  const int myScore = (int)(myCw >> 1);
  const uint64_t keep = __ballot(myKeep);
  long long bestInside = -1, bestOutside = -1;  // signer << 8 | slot of the entry
  {
    const int top = wave_reduce_max_i32(myInside ? myScore : 0);
    if (top > 0) {
      const uint64_t hm = __ballot(myInside && myScore == top);
      bestInside = (long long)lane_bcast((uint32_t)myEnt, __ffsll((unsigned long long)hm) - 1);
    }
    const uint64_t om = __ballot(myKeep && !myInside);
    if (bestInside < 0 && om) {
      const int lo = wave_reduce_min_i32(myKeep && !myInside ? myRank : INT32_MAX);
      const uint64_t hm = __ballot(myKeep && !myInside && myRank == lo);
      bestOutside = (long long)lane_bcast((uint32_t)myEnt, __ffsll((unsigned long long)hm) - 1);
    }
  }
  const int kept = __popcll(keep);
  unsigned long long relMask = 0;
  if (kept != len) {  // replaceToVerifyAgg :636-646
    const int newPos = __popcll(keep & lanes_lt());
    const bool mineKept = lane < len && ((keep >> lane) & 1ULL);
    if (mineKept && newPos != lane) ent[newPos] = myEnt;
    const uint32_t key = h_pend_word(l, mySlot);
    const bool held = pend.x == key || pend.y == key || pend.z == key || pend.w == key;
    const uint64_t rel = __ballot(lane < len && !mineKept && !held);
    for (uint64_t m = rel; m; m &= m - 1) relMask |= 1ULL << lane_bcast((uint32_t)mySlot, __ffsll((unsigned long long)m) - 1);
  }
  uint32_t sum0 = 0, sum1 = 0;
  if (!ATK && kept > 0) {  // the level's summary over the entries that stay listed (h_summary_has_candidate)
    sum0 = (uint32_t)wave_reduce_min_i32(myKeep ? myRank : INT32_MAX) | (__ballot(myKeep && myScore > 0) ? 0x80000000u : 0u);
    sum1 = (uint32_t)wave_reduce_max_i32(myKeep ? myRank : 0);
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    if (!ATK && newValid != valid0) qr[H_QVALID] = newValid;
    h_item_finish(s, node, l, qr, qh, len, kept, relMask, bestInside >= 0 ? bestInside : bestOutside, sum0, sum1);
  }
  KPROF_MARK(d.g, 23);  // list curation, candidate
  return bestInside >= 0 ? bestInside : bestOutside;
}

// A1: bestToVerify (:570-634) of one (runner, level) item with something to evaluate: curates the level's list, records
// its candidate and summary. k_handel_a1c runs both kinds of items in one launch — the narrow levels' by groups of eight
// lanes, the wide levels' one WAVEFRONT each —; k_handel_a1w (wave items only) is what an attack's run uses. (As two launches
// the narrow half is one long chain per item — 44 us by itself — which the one launch hides behind the wave items: profiles/r13h_*.)
// ---------------- one GROUP of eight lanes per item: the narrow levels (blocks of <= 16 words), two words a lane ---------
// One lane per item walked a block of up to 16 words two words a load, one entry after the other: a chain of up to a dozen
// dependent round trips that made the lane half of k_handel_a1 the phase's longest (44 us of an ordinary ms for ~ 5 k items an
// engine, profiles/r13g_*). Eight lanes hold the block's words side by side (lane j: words 2j, 2j + 1), lane j owns entry j of
// the list, the sums meet by three DPP steps inside the half-row: an item is three round trips whatever its level.
#if defined(__HIPCC__) && !defined(WG_NO_DPP)
// xor 1, xor 2 inside a quad, then row_half_mirror (lane i <-> 7 - i of its half-row: the other quad's total)
template <class OP>
__device__ __forceinline__ uint32_t group8_reduce32(uint32_t v, uint32_t id) {
  v = OP::f(v, dpp_mov<0xb1, 0xf>(id, v));
  v = OP::f(v, dpp_mov<0x4e, 0xf>(id, v));
  v = OP::f(v, dpp_mov<0x141, 0xf>(id, v));
  return v;
}
__device__ __forceinline__ uint64_t group8_sum64(uint64_t v) {
  v = v + dpp_mov64<0xb1, 0xf>(0ull, v);
  v = v + dpp_mov64<0x4e, 0xf>(0ull, v);
  v = v + dpp_mov64<0x141, 0xf>(0ull, v);
  return v;
}
__device__ __forceinline__ uint64_t group8_or64(uint64_t v) {
  v = v | dpp_mov64<0xb1, 0xf>(0ull, v);
  v = v | dpp_mov64<0x4e, 0xf>(0ull, v);
  v = v | dpp_mov64<0x141, 0xf>(0ull, v);
  return v;
}
#else
template <class OP>
__device__ __forceinline__ uint32_t group8_reduce32(uint32_t v, uint32_t) {
  for (int o = 1; o < 8; o <<= 1) v = OP::f(v, (uint32_t)__shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ uint64_t group8_sum64(uint64_t v) {
  for (int o = 1; o < 8; o <<= 1) v += shfl64(v, WG_LANE ^ o);
  return v;
}
__device__ __forceinline__ uint64_t group8_or64(uint64_t v) {
  for (int o = 1; o < 8; o <<= 1) v |= shfl64(v, WG_LANE ^ o);
  return v;
}
#endif
__device__ __forceinline__ int32_t group8_min_i32(int32_t v) {
  return (int32_t)(group8_reduce32<OpMin>((uint32_t)v ^ 0x80000000u, 0xFFFFFFFFu) ^ 0x80000000u);
}
__device__ __forceinline__ int32_t group8_max_i32(int32_t v) {
  return (int32_t)(group8_reduce32<OpMax>((uint32_t)v ^ 0x80000000u, 0u) ^ 0x80000000u);
}

__device__ __forceinline__ void h_a1_group_items(const EngineDev& d, const HandelState& s, uint32_t block, uint32_t nBlocks) {
  KPROF_DECL;
  const int lane = WG_LANE, j = lane & 7, gbase = lane & ~7;
  const uint32_t nItems = s.itemCount[0];
  const uint32_t stride = nBlocks * (blockDim.x >> 3);
  // (every lane of the wavefront stays in the loop while any group has an item: the DPP steps read all lanes)
  for (uint32_t q = block * (blockDim.x >> 3) + (threadIdx.x >> 3); __ballot(q < nItems); q += stride) {
    const bool have = q < nItems;
    const uint32_t it = have ? s.itemsLane[q] : 0u;
    const int32_t node = (int32_t)(it & 0x00FFFFFFu);
    const int l = have ? (int)(it >> 24) : 1;
    const uint32_t WG_G* hdr = h_hdr(s, node);
    const Lv v = sib_view(node, l);
    const uint64_t WG_G* ti = h_row(s, node, HK_TI, l);
    const uint64_t WG_G* la = h_row(s, node, HK_LA, l);
    const uint64_t WG_G* vi = h_row(s, node, HK_VI, l);
    uint64_t WG_G* qr = h_qrec(s, node, l);
    uint64_t WG_G* ent = qr + H_QENT;
    uint32_t WG_G* cache = s.qcache + ((size_t)node * s.L + l) * (size_t)s.QC;
    // ---- everything the item's address alone decides, before the first use
    int window = 0;
    U4 lvA;
    lvA.x = lvA.y = lvA.z = lvA.w = 0;
    HQHead qh;
    qh.len = qh.used = 0;
    uint64_t valid0 = 0, myEnt = ~0ULL;
    uint32_t c8 = 0;
    V2 tiw, viw, law;
    tiw.x = tiw.y = viw.x = viw.y = law.x = law.y = 0;
    if (have) {
      window = (int)hdr[HH_WINDOW];
      lvA = gld((const U4 WG_G*)h_lv(s, node, HP_POS, l));
      qh = gld((const HQHead WG_G*)qr);
      valid0 = qr[H_QVALID];
      myEnt = ent[j];      // entry j of the list's first eight
      c8 = cache[j];       // the cached evaluation of SLOT j (slots are taken lowest first)
      if (v.nw == 1) {
        if (j == 0) {
          viw.x = vi[0] & v.mask;
          law.x = la[0] & v.mask;
        }
      } else if (2 * j < v.nw) {
        viw = gld((const V2 WG_G*)(vi + 2 * j));
        law = gld((const V2 WG_G*)(la + 2 * j));
      }
      tiw.x = law.x | viw.x;  // totalIncoming = lastAggVerified | verifiedIndSignatures (the row is not read)
      tiw.y = law.y | viw.y;
    }
    const int len = have ? (int)qh.len : 0, curSize = (int)lvA.y, cLA = (int)lvA.z;
    uint64_t newValid = valid0;
    // the evaluations of a chunk of eight entries (lane j: entry 8c + j): cached, or the signature against the three sets
    auto evaluate = [&](bool mine, uint64_t e, uint32_t cw0, bool valid) -> uint32_t {
      uint32_t myCw = valid ? cw0 : 0u;
      uint32_t inv8 = (uint32_t)((__ballot(mine && !valid) >> gbase) & 0xFFu);
      while (__ballot(inv8 != 0)) {
        const bool act = inv8 != 0;
        const int pos = act ? __ffs(inv8) - 1 : 0;
        inv8 &= inv8 - 1;
        const int slot = (int)((uint32_t)__shfl((int)(uint32_t)e, gbase | pos, 64) & 0xFFu);
        const uint64_t WG_G* sig = h_sig_ptr(s, node, l, slot);
        V2 sg;
        sg.x = sg.y = 0;
        if (act) {
          if (v.nw == 1) {
            if (j == 0) sg.x = sig[0];
          } else if (2 * j < v.nw) {
            sg = gld((const V2 WG_G*)(sig + 2 * j));
          }
        }
        uint64_t a = (uint64_t)(__popcll(sg.x | tiw.x | viw.x) + __popcll(sg.y | tiw.y | viw.y)) |
                     ((uint64_t)(__popcll(sg.x | viw.x) + __popcll(sg.y | viw.y)) << 21) |
                     ((uint64_t)(__popcll(sg.x) + __popcll(sg.y)) << 42);
        uint64_t b = (uint64_t)(((sg.x & tiw.x) | (sg.y & tiw.y)) != 0) | ((uint64_t)(((sg.x & law.x) | (sg.y & law.y)) != 0) << 21);
        a = group8_sum64(a);
        b = group8_sum64(b);
        const int u1 = (int)(a & 0x1FFFFF), u2 = (int)((a >> 21) & 0x1FFFFF), cs = (int)((a >> 42) & 0x1FFFFF);
        const bool iTI = (b & 0x1FFFFF) != 0, iLA = ((b >> 21) & 0x1FFFFF) != 0;
        const uint32_t cw = h_eval_word(u1, u2, cs, iTI, iLA, curSize, cLA, v.size);
        if (act) {
          if (j == pos) {
            myCw = cw;
            __hip_atomic_store(cache + slot, cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          newValid |= 1ULL << slot;
        }
      }
      return myCw;
    };
    // ---- first chunk: entries 0 .. 7 (all there is, but for the rare long list)
    const bool mine0 = have && j < len;
    const int slot0 = mine0 ? (int)(myEnt & 0xFF) : 0;
    const bool valid00 = mine0 && ((valid0 >> slot0) & 1ULL);
    const uint32_t c8s = (uint32_t)__shfl((int)c8, gbase | (slot0 & 7), 64);
    uint32_t cw0 = 0;
    if (valid00) cw0 = slot0 < 8 ? c8s : cache[slot0];
    uint32_t myCw = evaluate(mine0, myEnt, cw0, valid00);
    int myRank = mine0 ? (int)(uint32_t)(myEnt >> 32) : INT32_MAX;
    int windowIndex = group8_min_i32(myRank);  // Collections.min(rank)
    const bool isLong = len > 8;
    if (__ballot(isLong)) {  // (rare) the other chunks: their evaluations go to the cache, their ranks into the minimum
      for (int c = 1; __ballot(isLong && 8 * c < len); c++) {
        const bool mc = isLong && 8 * c + j < len;
        const uint64_t e = mc ? ent[8 * c + j] : ~0ULL;
        const int sl = mc ? (int)(e & 0xFF) : 0;
        const bool vc = mc && ((valid0 >> sl) & 1ULL);
        (void)evaluate(mc, e, 0u, vc);
        windowIndex = min(windowIndex, group8_min_i32(mc ? (int)(uint32_t)(e >> 32) : INT32_MAX));
      }
      __threadfence_block();
    }
    // ---- the choice (as h_best_wave), a chunk at a time with the best so far carried along: the curated list, the best
    // inside / outside the window, the level's summary
    long long bestInside = -1, bestOutside = -1;
    int bestScore = 0, bestOutsideRank = 0, kept = 0, sumMin = INT32_MAX, sumMax = 0;
    bool sumPos = false;
    uint64_t relMask = 0;
    U4 pend;
    pend.x = pend.y = pend.z = pend.w = 0;
    bool pendLoaded = false;
    for (int c = 0; __ballot(have && 8 * c < len); c++) {
      const bool mc = have && 8 * c + j < len;
      if (c > 0) {  // (long lists: entry and evaluation from memory; every entry has one by now)
        myEnt = mc ? ent[8 * c + j] : ~0ULL;
        myRank = mc ? (int)(uint32_t)(myEnt >> 32) : INT32_MAX;
        myCw = mc ? __hip_atomic_load(cache + (myEnt & 0xFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      }
      const int mySlot = mc ? (int)(myEnt & 0xFF) : 0;
      const bool myKeep = mc && (myCw & 1u);
      const bool myInside = myKeep && h_in_window(myRank, windowIndex, window);
      const int myScore = (int)(myCw >> 1);
      const uint32_t keep8 = (uint32_t)((__ballot(myKeep) >> gbase) & 0xFFu);
      const int top = group8_max_i32(myInside ? myScore : 0);
      if (top > bestScore) {  // the FIRST entry with the strictly greatest positive score
        const uint32_t hm = (uint32_t)((__ballot(myInside && myScore == top) >> gbase) & 0xFFu);
        bestScore = top;
        bestInside = (long long)(uint32_t)__shfl((int)(uint32_t)myEnt, gbase | ((hm ? __ffs(hm) : 1) - 1), 64);
      }
      const uint32_t om = (uint32_t)((__ballot(myKeep && !myInside) >> gbase) & 0xFFu);
      if (om) {  // the FIRST entry with the smallest rank
        const int lo = group8_min_i32(myKeep && !myInside ? myRank : INT32_MAX);
        if (bestOutside < 0 || lo < bestOutsideRank) {
          const uint32_t hm = (uint32_t)((__ballot(myKeep && !myInside && myRank == lo) >> gbase) & 0xFFu);
          bestOutsideRank = lo;
          bestOutside = (long long)(uint32_t)__shfl((int)(uint32_t)myEnt, gbase | ((hm ? __ffs(hm) : 1) - 1), 64);
        }
      }
      const int chunkLen = min(8, len - 8 * c), keptC = __popc(keep8);
      if (__ballot(mc && (keptC != chunkLen || kept != 8 * c))) {  // replaceToVerifyAgg :636-646 (entries move down)
        const bool cur = have && (keptC != chunkLen || kept != 8 * c);
        if (cur && !pendLoaded) {
          pend = gld((const U4 WG_G*)(hdr + HH_PEND));
          pendLoaded = true;
        }
        const int newPos = kept + __popc(keep8 & ((1u << j) - 1u));
        if (cur && myKeep && newPos != 8 * c + j) ent[newPos] = myEnt;
        const uint32_t key = h_pend_word(l, mySlot);
        const bool held = pend.x == key || pend.y == key || pend.z == key || pend.w == key;
        relMask |= group8_or64(cur && mc && !myKeep && !held ? 1ULL << mySlot : 0ULL);
      }
      kept += keptC;
      sumMin = min(sumMin, group8_min_i32(myKeep ? myRank : INT32_MAX));
      sumMax = max(sumMax, group8_max_i32(myKeep ? myRank : 0));
      sumPos = sumPos || ((__ballot(myKeep && myScore > 0) >> gbase) & 0xFFu) != 0;
    }
    if (have && j == 0) {
      if (newValid != valid0) qr[H_QVALID] = newValid;
      h_item_finish(s, node, l, qr, qh, len, kept, relMask, bestInside >= 0 ? bestInside : bestOutside,
                    (uint32_t)sumMin | (sumPos ? 0x80000000u : 0u), (uint32_t)sumMax);
#ifdef WG_KPROF
      atomicAdd(F(&d.g->kprofBuf[KPROF_WAVE + 28]), 1ULL);
      atomicAdd(F(&d.g->kprofBuf[KPROF_WAVE + 29]), (unsigned long long)len);
      atomicAdd(F(&d.g->kprofBuf[KPROF_WAVE + 17]), (unsigned long long)__popcll(newValid & ~valid0));
#endif
    }
  }
  KPROF_MARK(d.g, 1);   // a group-part wavefront, start to end
  KPROF_COUNT(d.g, 2);  // ... how many
}

// ---------------- one wavefront per item: lanes = 64-bit words of the level's block ----------------
template <bool ATK>
__device__ __forceinline__ void h_a1_wave_items(const EngineDev& d, const HandelState& s, uint32_t block, uint32_t nBlocks) {
  const uint32_t wave = (block * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (nBlocks * blockDim.x) >> 6;
  const uint32_t nItems = s.itemCount[1];
  for (uint32_t q = wave; q < nItems; q += nWaves) {
    const uint32_t it = WG_READFIRST(s.itemsWave[q]);  // (wave-uniform: the item's addresses live in SGPRs)
    h_best_wave<ATK>(d, s, (int32_t)(it & 0x00FFFFFFu), (int)((it >> 24) & 31u), !ATK && (it >> 31) != 0);
  }
}
template <int WPE, bool ATK>
__global__ void __launch_bounds__(256, WPE) k_handel_a1w(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  h_a1_wave_items<ATK>(d, stab[wgBy], wgBx, wgGx);
}
// both in ONE launch: the first half of the blocks the narrow levels' items (groups of eight lanes), the others the wave
// items — the two kinds of chains in flight together, at the group half's register count
template <int WPE>
__global__ void __launch_bounds__(256, WPE) k_handel_a1c(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  const uint32_t laneBlocks = wgGx >= 4 ? wgGx / 2u : 1;
  if (wgBx < laneBlocks)
    h_a1_group_items(d, s, wgBx, laneBlocks);
  else
    h_a1_wave_items<false>(d, s, wgBx - laneBlocks, wgGx - laneBlocks);
}

// scan over nodes: ordinal of each node that draws (checkSigs draws iff some level has a candidate),
// and the draw itself — chooseBestFromLevels: rd.nextInt(byLevels.size()) (:788-790) — by jump-ahead
// assuming no earlier nextInt(bound) rejection; a rejection anywhere is flagged and re-walked in A2.
// byLevels.size() of node i at this edge: its own mask's bits, or (sharded) the byte the owner's shard reported
__device__ __forceinline__ int32_t h_cand_count(const HandelState& s, uint32_t i) {
  if (s.xcand) return (int32_t)(((uint32_t)s.xcand[i >> 2] >> ((i & 3u) * 8u)) & 0xFFu);
  return (int32_t)__popc(s.candMask[i]);
}
// sharded: the owners' counts into the exchange image (zeros for the nodes of other shards)
__global__ void __launch_bounds__(256) k_handel_cand_pack(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  const uint32_t nw = ((uint32_t)s.N + 3u) >> 2;
  for (uint32_t j = wgBx * blockDim.x + threadIdx.x; j < nw; j += wgGx * blockDim.x) {
    uint32_t w = 0;
    for (uint32_t q = 0; q < 4; q++) {
      const uint32_t i = 4 * j + q;
      if (i >= (uint32_t)s.lo && i < (uint32_t)s.hi) w |= (uint32_t)__popc(s.candMask[i]) << (8 * q);
    }
    s.xcand[j] = (int32_t)w;
  }
}
struct CondF {
  typedef HandelState Aux;
  const EngineDev& d;
  const HandelState& s;
  __device__ CondF(const EngineDev& d_, const Aux* a) : d(d_), s(*a) {}
  __device__ uint32_t count() const { return (uint32_t)s.N; }
  __device__ uint64_t value(uint32_t i) const { return h_cand_count(s, i) != 0; }
  __device__ void tally(uint32_t, uint32_t) const {}
  __device__ void total(uint64_t tot) const {
    if (d.g->nOutKeep + (uint32_t)tot > d.maxOut) {  // (the drain's outbox is still in fin / arr: the edge's records follow it)
      set_err(d.g, ERR_OUTBOX);
      tot = 0;
    }
    d.g->nOut = (uint32_t)tot;  // one registerTask per drawing node
    d.g->nDraws = (uint32_t)tot;
  }
  __device__ void write(uint32_t i, uint64_t excl, bool valid) const {
    const int32_t nc = h_cand_count(s, i);
    if (!valid || nc == 0) return;
    s.condList[(uint32_t)excl] = i;
    uint64_t st = lcg_skip(d.g->rng, excl);
    int consumed;
    s.drawVal[i] = lcg_next_int_bounded(st, nc, &consumed);
    if (consumed != 1) d.g->rejectSeen = 1;
  }
};

// HiddenByzantine.attack (P/Handel.java:861-916), between the draw among the levels and the rest of checkSigs: for a node
// whose drawn candidate is of the LAST level, the byzantine peer of best rank that is not in totalIncoming yet plants a
// valid one-signer signature in that level's list if it outranks the candidate, and the level's bestToVerify runs again
// (firstByzantine :844-858, the planted entry remembered in `last` until it is verified or pruned). One wavefront per
// drawing node; k_handel_cond_a2 then reads the level's candidate as this kernel left it.
__global__ void __launch_bounds__(256) k_handel_hidden(const EngineDev* __restrict__ tab, const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  const int lane = WG_LANE;
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t n = d.g->nOut;
  if (d.g->rejectSeen) {  // (the draws are re-walked by k_handel_cond_a2 only: p < 2^-30 per draw)
    if (wave == 0 && lane == 0 && n) set_err(d.g, ERR_PROTOCOL);
    return;
  }
  const int top = s.L - 1;
  if (top < 1) return;
  for (uint32_t j = wave; j < n; j += nWaves) {
    const int32_t node = (int32_t)s.condList[j];
    uint32_t cm = s.candMask[node];
    for (int q = s.drawVal[node]; q > 0; q--) cm &= cm - 1;
    if (__ffs(cm) - 1 != top) continue;  // :813 best.level == levels.size() - 1
    uint32_t WG_G* h = h_hdr(s, node);
    if (h[HH_HB_NOBYZ]) continue;
    const uint32_t who = *h_lv(s, node, HP_CAND, top);  // currentBest: signer << 8 | slot
    const uint32_t last = h[HH_HB_LAST];
    __builtin_amdgcn_wave_barrier();
    if (last && last - 1 == who) {  // a previous attack finally worked
      if (lane == 0) h[HH_HB_LAST] = 0;
      continue;
    }
    uint64_t WG_G* qr = h_qrec(s, node, top);
    uint64_t WG_G* ent = qr + H_QENT;
    const int len = (int)qr[0];
    const unsigned long long used = qr[1];
    const uint64_t myEnt = lane < len ? ent[lane] : ~0ULL;
    const Lv v = sib_view(node, top);
    const uint64_t WG_G* ti = h_row(s, node, HK_TI, top);
    if (last) {
      if (__ballot(lane < len && (uint32_t)myEnt == last - 1)) continue;  // still listed: nothing new
      const int32_t lf = (int32_t)((last - 1) >> 8);
      if (!((ld_coherent(ti + ((lf >> 6) - v.bw)) >> (lf & 63)) & 1ULL)) {
        if (lane == 0) set_err(d.g, ERR_PROTOCOL);  // "byz signature pruned!"
        continue;
      }
      if (lane == 0) h[HH_HB_LAST] = 0;
    }
    // firstByzantine: the down peer of the level with the smallest rank (the first of them in list order) not in totalIncoming
    const int size = 1 << (top - 1);
    const size_t peers0 = (size_t)node * (s.N - 1) + (size - 1);
    unsigned long long key = ~0ULL;
    for (int i0 = 0; i0 < size; i0 += 64) {
      const int i = i0 + lane;
      if (i < size) {
        const int32_t p = h_peer(s, peers0 + i);
        if (d.nodes.down[p] && !((ld_coherent(ti + ((p >> 6) - v.bw)) >> (p & 63)) & 1ULL)) {
          const unsigned long long k2 = ((unsigned long long)(uint32_t)s.ranks[(size_t)node * s.N + p] << 32) | (uint32_t)i;
          key = k2 < key ? k2 : key;
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = shfl64(key, lane ^ o);
      key = other < key ? other : key;
    }
    key = lane_bcast64(key, 0);
    if (key == ~0ULL) {
      if (lane == 0) h[HH_HB_NOBYZ] = 1;
      continue;
    }
    const int fbRank = (int)(uint32_t)(key >> 32);
    const int32_t fb = h_peer(s, peers0 + (uint32_t)key);
    const uint64_t hitM = __ballot(lane < len && (uint32_t)myEnt == who);
    if (!hitM) {
      if (lane == 0) set_err(d.g, ERR_PROTOCOL);
      continue;
    }
    const int bestRank = (int)lane_bcast((uint32_t)(myEnt >> 32), __ffsll((unsigned long long)hitM) - 1);
    if (fbRank >= bestRank) continue;  // we can't improve it, we're too far
    const int qc = h_qcap(s, top);
    const unsigned long long capMask = qc >= 64 ? ~0ULL : ((1ULL << qc) - 1ULL);
    const unsigned long long freeM = ~used & capMask;
    if (freeM == 0 || len >= 64) {
      if (lane == 0) set_err(d.g, ERR_QUEUE_CAP);
      continue;
    }
    const int slot = __ffsll(freeM) - 1;
    uint64_t WG_G* dst = h_sig_ptr(s, node, top, slot);
    H_FOR_WORDS(v, jw) dst[jw] = jw == (fb >> 6) - v.bw ? 1ULL << (fb & 63) : 0ULL;  // sig = { fb }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      ent[len] = h_entry(fbRank, fb, slot);
      HQHead nh;
      nh.len = (uint64_t)(len + 1);
      nh.used = used | (1ULL << slot);
      gst((HQHead WG_G*)qr, nh);
      qr[s.qBad] &= ~(1ULL << slot);  // badSig = false
      atomicAdd(F(h + HH_SIGQ), 1u);
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    const long long newBest = h_best_wave<true>(d, s, node, top);  // SigToVerify newBest = l.bestToVerify()
    const uint32_t planted = ((uint32_t)fb << 8) | (uint32_t)slot;
    if (newBest < 0) {
      if (lane == 0) set_err(d.g, ERR_PROTOCOL);  // (the reference dereferences null at :819)
      continue;
    }
    if ((uint32_t)newBest != planted && lane == 0) h[HH_HB_LAST] = planted + 1;
    __builtin_amdgcn_wave_barrier();
  }
}

// A2: the rest of checkSigs (:816-836) for the drawn candidate, one lane per drawing node.
// SH (sharded engine): a drawing node is handled by its owner, the task record goes to the exchange image.
template <bool SH, bool ATK>
__global__ void __launch_bounds__(256) k_handel_cond_a2(const EngineDev* __restrict__ tab,
                                                        const HandelState* __restrict__ stab) {
  WG_ENGINE(tab);
  const HandelState& s = stab[wgBy];
  const uint32_t n = d.g->nOut;
  const int32_t t = d.g->now;
  const bool rejected = d.g->rejectSeen != 0;
  const uint32_t D = (uint32_t)d.horizon;
  const uint32_t stride = wgGx * blockDim.x;
  if (wgBx == 0 && threadIdx.x < 2) s.itemCount[threadIdx.x] = 0;  // for the next edge's k_handel_cond_pre
  for (uint32_t j0 = wgBx * blockDim.x; j0 < n; j0 += stride) {
    const uint32_t j = j0 + threadIdx.x;
    uint32_t histKey = 0xFFFFFFFFu;
    if (j < n) {
      const int32_t node = (int32_t)s.condList[j];
      int k = s.drawVal[node];
      const bool owned = !SH || shard_owns(d, node);
      if (rejected && (owned || j + 1 == n)) {  // a nextInt(bound) rejection shifted the stream: walk it serially up to this draw (rare)
        uint64_t st = d.g->rng;
        uint32_t total = 0;
        for (uint32_t q = 0; q <= j; q++) {
          int consumed;
          k = lcg_next_int_bounded(st, h_cand_count(s, s.condList[q]), &consumed);
          total += (uint32_t)consumed;
        }
        if (j + 1 == n) d.g->nDraws = total;
      }
      if (!owned) {
        for (int q = 0; q < 5; q++) d.xbuf[(size_t)j * 5 + q] = 0;
        continue;
      }
      uint32_t WG_G* h = h_hdr(s, node);
      // byLevels is in level order: the k-th level with a candidate, its candidate's queue slot
      uint32_t cm = s.candMask[node];
      for (int q = 0; q < k; q++) cm &= cm - 1;
      const int l = __ffs(cm) - 1;
      // the level's candidate, signer << 8 | queue slot: recorded by this edge's item (h_item_finish), or — a clean level,
      // which had no item — picked now from the cached evaluations
      uint32_t who;
      if ((s.cleanMask[node] >> l) & 1u) {
        const long long pick = h_pick_cached(s, node, l, (int)h[HH_WINDOW]);
        if (pick < 0) set_err(d.g, ERR_PROTOCOL);
        who = (uint32_t)pick;
      } else {
        who = *h_lv(s, node, HP_CAND, l);
      }
      const int slot = (int)(who & 0xFFu);
      const int32_t from = (int32_t)(who >> 8);
      // currWindowSize = min(window.newSize(cur, correct = true), l.size)  (:821-822, ScoringExp :192-200)
      int w = (int)h[HH_WINDOW] * 2;
      if (ATK && ((h_qrec(s, node, l)[s.qBad] >> slot) & 1ULL)) w = (int)h[HH_WINDOW] / 4;  // newSize(cur, !best.badSig): floor(cur / 4)
      if (w > s.p.windowMaximum) w = s.p.windowMaximum;
      if (w < s.p.windowMinimum) w = s.p.windowMinimum;
      h[HH_WINDOW] = (uint32_t)min(w, 1 << (l - 1));
      // receptionRanks[best.from] += nodeCount, saturating (:825-828)
      if (s.ranks) {
        int32_t WG_G* rk = s.ranks + (size_t)node * s.N + from;
        int32_t nr = (int32_t)((uint32_t)*rk + (uint32_t)s.N);
        *rk = nr < 0 ? INT32_MAX : nr;
        if (nr < 0) atomicOr(&d.g->notes, NOTE_RANKS_SATURATED);
      } else {
        h_bump(d, s, (int32_t)node, l, from);
      }
      h[HH_SIGCHK]++;
      int pe = -1;
      for (int q = 0; q < H_PEND; q++)
        if (!(h[HH_PEND + q] & 0x80000000u)) {
          pe = q;
          break;
        }
      if (pe < 0) {
        set_err(d.g, ERR_PENDING);
        pe = 0;
      }
      h[HH_PEND + pe] = h_pend_word(l, slot);
      h[HH_PENDFROM + pe] = (uint32_t)from;
      // registerTask(updateVerifiedSignatures(best), time + nodePairingTime, this)
      const int32_t arrival = t + (int32_t)h[HH_PAIR];
      const Rec fin = make_rec(K_TASK, node, (uint32_t)node, H_TASK_UPDATE, h_update_arg(pe, l, slot, from));
      const bool ok = arrival - t < d.horizon - 1;  // see Engine::run_ms on host-held envelopes
      if (!ok) set_err(d.g, ERR_HORIZON);
      if (SH) {
        int32_t WG_G* x = d.xbuf + (size_t)j * 5;
        x[0] = (int32_t)fin.w0;
        x[1] = (int32_t)fin.w1;
        x[2] = (int32_t)fin.w2;
        x[3] = (int32_t)fin.w3;
        x[4] = ok ? arrival + 1 : 0;
        continue;
      }
      const uint32_t jp = d.g->nOutKeep + j;  // (behind the drain's records, if the drain left them to this phase's append)
      d.fin[jp] = fin;
      d.arr[jp] = ok ? arrival : -1;
      if (ok) histKey = (jp / TILE) * D + ((uint32_t)arrival & (D - 1));
    }
    if (SH) continue;  // (k_shard_unpack builds the tile histograms from the summed image)
    // per-tile arrival histogram of the multisplit; pairing times are nearly uniform, so aggregate equal
    // keys inside the wavefront before touching memory
    uint64_t todo = __ballot(histKey != 0xFFFFFFFFu);
    while (todo) {
      const int leader = __ffsll((unsigned long long)todo) - 1;
      const uint32_t key = lane_bcast(histKey, leader);
      const uint64_t m = __ballot(histKey == key) & todo;
      if ((int)WG_LANE == leader) atomicAdd(&d.tileHist[key], (uint32_t)__popcll(m));
      todo &= ~m;
    }
  }
}

// ---- init() on the device: the reception ranks (P/Handel.java:966-989: one list of all nodes, Collections.shuffle'd
// again for every node in id order; node n's receptionRanks[x] = the position of x in the list after its shuffle) ----
// rd is one stream over the N * (N - 1) nextInt(i) draws, and the list carries over from node to node. Both chains are cut:
//  * where a node's draws start depends on the rejected draws before it (java.util.Random.nextInt's loop; about N^3 / 2^33
//    of them). A draw can only be rejected when its 31 bits are >= 2^31 - N, whatever its bound: k_handel_init_scan lists
//    those stream positions (about N^3 / 2^31), the host walks the short list in order (position - rejections so far =
//    the draw, hence its bound) and hands every node its start;
//  * a shuffle moves POSITIONS: applied to the identity it gives p_n with list_n[j] = list_(n-1)[p_n[j]]. k_handel_init_perm
//    builds p_n for all nodes at once (one lane a node, in the node's own row of `ranks`), k_handel_init_chain composes them
//    in node order with the list in LDS and scatters row n = the inverse of list_n over p_n's row — composition is
//    associative, so the node order is cut into chunks that run side by side (see there).
__global__ void __launch_bounds__(256) k_handel_init_scan(uint64_t rng0, unsigned long long total, uint32_t N,
                                                          unsigned long long* __restrict__ cand, uint32_t* __restrict__ candCount,
                                                          uint32_t cap) {
  const unsigned long long CH = 2048;
  for (unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; c * CH < total;
       c += (unsigned long long)gridDim.x * blockDim.x) {
    uint64_t st = lcg_skip(rng0, c * CH);
    const unsigned long long end = c * CH + CH < total ? c * CH + CH : total;
    for (unsigned long long pos = c * CH; pos < end; pos++) {
      st = lcg_step(st);
      const uint32_t bits = (uint32_t)(st >> 17);  // next(31)
      if (bits >= 0x80000000u - N) {
        const uint32_t i = atomicAdd(candCount, 1u);
        if (i < cap) cand[i] = (pos << 20) | (0x7FFFFFFFu - bits);  // (the distance from 2^31 - 1: below N <= 2^20; pos < 2^44)
      }
    }
  }
}
__global__ void __launch_bounds__(64) k_handel_init_perm(HandelState s, const unsigned long long* __restrict__ offs, uint64_t rng0,
                                                         uint32_t* __restrict__ bad) {
  const int N = s.N;
  const int n0 = blockIdx.x * 64;
  for (int r = 0; r < 64 && n0 + r < N; r++) {  // the identity, a row at a time by the whole wavefront
    int32_t* row = s.ranks + (size_t)(n0 + r) * N;
    for (int j = threadIdx.x; j < N; j += 64) row[j] = j;
  }
  __syncthreads();
  const int n = n0 + (int)threadIdx.x;
  if (n >= N) return;
  int32_t* row = s.ranks + (size_t)n * N;
  uint64_t st = lcg_skip(rng0, offs[n]);
  unsigned long long drawn = 0;
  for (int32_t i = N; i > 1; i--) {  // Collections.shuffle: swap(i - 1, rd.nextInt(i))
    int consumed;
    const int32_t j = lcg_next_int_bounded(st, i, &consumed);
    drawn += (unsigned long long)consumed;
    const int32_t a = row[i - 1], b = row[j];
    row[i - 1] = b;
    row[j] = a;
  }
  if (drawn != offs[n + 1] - offs[n]) atomicOr(bad, 1u);  // (the host's walk of the candidates and the draws disagree)
}
// The list (ids < 65 536) in LDS, thread t owns positions t + k * blockDim, N == E * blockDim. The node order is cut
// into chunks of B nodes, one workgroup each, in three launches:
//   ROWS == false  compose the chunk's B shuffles from the identity: its net move q_c (list after = list before[q_c[j]])
//   k_handel_init_chain_starts (one workgroup)  the list every chunk starts from: start_0 = identity, start_(c+1) = start_c[q_c[j]]
//   ROWS == true   the chunk again from its start list; row n = the inverse of the list after node n, scattered over p_n
template <int E, bool ROWS>
__global__ void __launch_bounds__(1024) k_handel_init_chain(HandelState s, int B, uint16_t* __restrict__ net,
                                                            const uint16_t* __restrict__ starts) {
  WG_DYN_LDS(uint16_t, lst);  // [N]
  const int N = s.N, T = E > 1 ? 1024 : (int)blockDim.x, t = (int)threadIdx.x;
  constexpr bool AHEAD = E <= 16;  // (the next node's row in registers while this one is composed; more would spill)
  const int n0 = (int)blockIdx.x * B;
  uint32_t p[E], pn[AHEAD ? E : 1];
#pragma unroll
  for (int k = 0; k < E; k++) lst[t + k * T] = ROWS ? starts[(size_t)blockIdx.x * N + t + k * T] : (uint16_t)(t + k * T);
  if (AHEAD) {
#pragma unroll
    for (int k = 0; k < E; k++) pn[k] = (uint32_t)s.ranks[(size_t)n0 * N + t + k * T];
  }
  __syncthreads();
  for (int n = n0; n < n0 + B; n++) {
    int32_t* row = s.ranks + (size_t)n * N;
    if (AHEAD) {
#pragma unroll
      for (int k = 0; k < E; k++) p[k] = pn[k];
      if (n + 1 < n0 + B) {
#pragma unroll
        for (int k = 0; k < E; k++) pn[k] = (uint32_t)row[N + t + k * T];
      }
    } else {
#pragma unroll
      for (int k = 0; k < E; k++) p[k] = (uint32_t)row[t + k * T];
    }
#pragma unroll
    for (int k = 0; k < E; k++) p[k] = lst[p[k]];
    __syncthreads();  // (every read of the old list and of row n is done)
#pragma unroll
    for (int k = 0; k < E; k++) {
      lst[t + k * T] = (uint16_t)p[k];
      if (ROWS) row[p[k]] = t + k * T;  // receptionRanks[id at position j] = j
    }
    __syncthreads();
  }
  if (!ROWS) {
#pragma unroll
    for (int k = 0; k < E; k++) net[(size_t)blockIdx.x * N + t + k * T] = lst[t + k * T];
  }
}
template <int E>
__global__ void __launch_bounds__(1024) k_handel_init_chain_starts(int N, int C, const uint16_t* __restrict__ net,
                                                                   uint16_t* __restrict__ starts) {
  WG_DYN_LDS(uint16_t, lst);
  const int T = E > 1 ? 1024 : (int)blockDim.x, t = (int)threadIdx.x;
  uint32_t p[E];
#pragma unroll
  for (int k = 0; k < E; k++) lst[t + k * T] = (uint16_t)(t + k * T);
  __syncthreads();
  for (int c = 0; c < C; c++) {
#pragma unroll
    for (int k = 0; k < E; k++) {
      starts[(size_t)c * N + t + k * T] = lst[t + k * T];
      p[k] = net[(size_t)c * N + t + k * T];
    }
#pragma unroll
    for (int k = 0; k < E; k++) p[k] = lst[p[k]];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < E; k++) lst[t + k * T] = (uint16_t)p[k];
    __syncthreads();
  }
}

// More than 65 536 nodes (or WG_INIT_BIG=1): ids no longer fit 16 bits and the list no longer fits LDS — the same three
// launches with the list in global memory (two buffers a chunk: a step reads one and writes the other; every word a
// thread reads was written by another thread of its workgroup before the barrier: L2-coherent loads)
__device__ __forceinline__ uint32_t ld_coherent32(const uint32_t WG_G* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool ROWS>
__global__ void __launch_bounds__(1024) k_handel_init_chain_big(HandelState s, int B, uint32_t* __restrict__ net,
                                                                const uint32_t* __restrict__ starts, uint32_t* __restrict__ work) {
  const int N = s.N, T = (int)blockDim.x, t = (int)threadIdx.x;
  const int n0 = (int)blockIdx.x * B;
  uint32_t* cur = work + (size_t)blockIdx.x * 2 * N;
  uint32_t* nxt = cur + N;
  for (int j = t; j < N; j += T) cur[j] = ROWS ? starts[(size_t)blockIdx.x * N + j] : (uint32_t)j;
  __syncthreads();
  for (int n = n0; n < n0 + B; n++) {
    int32_t* row = s.ranks + (size_t)n * N;
    for (int j = t; j < N; j += T) nxt[j] = ld_coherent32((const uint32_t WG_G*)cur + row[j]);
    __syncthreads();  // (every read of the old list and of row n is done)
    if (ROWS)
      for (int j = t; j < N; j += T) row[nxt[j]] = j;  // (own words of the new list) receptionRanks[id at position j] = j
    uint32_t* x = cur;
    cur = nxt;
    nxt = x;
  }
  if (!ROWS)
    for (int j = t; j < N; j += T) net[(size_t)blockIdx.x * N + j] = cur[j];
}
__global__ void __launch_bounds__(1024) k_handel_init_chain_starts_big(int N, int C, const uint32_t* __restrict__ net,
                                                                       uint32_t* __restrict__ starts, uint32_t* __restrict__ work) {
  const int T = (int)blockDim.x, t = (int)threadIdx.x;
  uint32_t* cur = work;
  uint32_t* nxt = work + N;
  for (int j = t; j < N; j += T) cur[j] = (uint32_t)j;
  __syncthreads();
  for (int c = 0; c < C; c++) {
    for (int j = t; j < N; j += T) {
      const uint32_t v = cur[j];  // (own word)
      starts[(size_t)c * N + j] = v;
      nxt[j] = ld_coherent32((const uint32_t WG_G*)cur + net[(size_t)c * N + j]);
    }
    __syncthreads();
    uint32_t* x = cur;
    cur = nxt;
    nxt = x;
  }
}

// ---- init() on the device: the emission lists (P/Handel.java:991-1013, buildEmissionList :510-522) -----------------
// For sender s and level l the receivers are the sibling block of 2^(l-1) ids; they are bucketed by the RECEIVER's
// receptionRanks[s], the buckets walked in rank order, a bucket of several receivers shuffled with the shared rd. The
// shuffles are the only sequential part — rd is one stream over (sender, level, bucket) in that order — and a bucket of
// g receivers draws exactly g - 1 times (a rejected draw, p ~ 1e-9, is flagged: the caller then builds the lists on the
// host). So: (A) every (sender, level) sorts its block by (rank, id) — one workgroup per sender, bitonic in LDS — writes
// the list unshuffled and counts its draws (elements - buckets); the host prefix-sums the counts in (sender, level)
// order; (B) every bucket of several receivers shuffles itself from the rd state jumped to its first draw.
// The receiver's rank of the sender is receptionRanks[receiver][sender]: a column of the uploaded matrix.
// (rank: the receiver v's reception rank of this sender — kept beside the id in the CARRIED form)
__device__ __forceinline__ void h_peer_store(const HandelState& s, size_t idx, int32_t v, uint32_t rank) {
  if (s.peersR)
    ((uint32_t WG_G*)(const uint32_t WG_G*)s.peersR)[idx] = (uint32_t)v | (rank << 16);
  else if (s.peers16)
    ((uint16_t WG_G*)(const uint16_t WG_G*)s.peers16)[idx] = (uint16_t)v;
  else
    ((int32_t WG_G*)(const int32_t WG_G*)s.peers32)[idx] = v;
}
// A level whose (rank, offset) no longer packs into 32 bits, or whose block no longer fits LDS (the last level of more
// than 65 536 nodes; `bigFrom`: levels from that one on), is sorted by counting instead: LDS histogram over rank >> shift,
// the elements scattered to `pairs` (global, [gridDim.x][N / 2] rank << 32 | offset) bin after bin, each bin — a few
// elements — put in (rank, offset) order by its thread.
__global__ void __launch_bounds__(1024) k_handel_init_sort(HandelState s, const uint8_t* __restrict__ down, uint32_t* __restrict__ cnt,
                                                           int bigFrom, int shift, unsigned long long* __restrict__ pairs) {
  WG_DYN_LDS(uint32_t, key);  // [N / 2] (rank << idBits | offset in the block): unique, so any sort is the stable one
  __shared__ uint32_t shGroups;
  __shared__ uint32_t shScan[16];
  const int N = s.N, L = s.L;
  for (int snd = blockIdx.x; snd < N; snd += gridDim.x) {
    if (down[snd]) {  // (a stopped node gets no levels' peers; its row stays as allocated and is never read)
      for (int l = threadIdx.x; l < L; l += blockDim.x) cnt[(size_t)snd * L + l] = 0;
      continue;
    }
    if (threadIdx.x == 0) cnt[(size_t)snd * L] = 0;
    for (int l = 1; l < L; l++) {
      const int m = 1 << (l - 1), base = ((snd >> (l - 1)) ^ 1) << (l - 1), idBits = l - 1;
      if (l >= bigFrom) {
        const int nbins = N >> shift, T = (int)blockDim.x, t = (int)threadIdx.x;
        unsigned long long* out = pairs + (size_t)blockIdx.x * (size_t)(N / 2);
        for (int b = t; b < nbins; b += T) key[b] = 0;
        if (t == 0) shGroups = 0;
        __syncthreads();
        for (int k = t; k < m; k += T) atomicAdd(&key[(uint32_t)s.ranks[(size_t)(base + k) * N + snd] >> shift], 1u);
        __syncthreads();
        {  // exclusive prefix over the bins: thread t owns the bins [t * per, (t + 1) * per)
          const int per = (nbins + T - 1) / T;
          uint32_t sum = 0;
          for (int b = t * per; b < (t + 1) * per && b < nbins; b++) sum += key[b];
          uint32_t total;
          uint32_t run = block_excl_scan32_1024(sum, shScan, &total);
          for (int b = t * per; b < (t + 1) * per && b < nbins; b++) {
            const uint32_t c0 = key[b];
            key[b] = run;
            run += c0;
          }
        }
        __syncthreads();
        for (int k = t; k < m; k += T) {
          const uint32_t rk = (uint32_t)s.ranks[(size_t)(base + k) * N + snd];
          out[atomicAdd(&key[rk >> shift], 1u)] = ((unsigned long long)rk << 32) | (uint32_t)k;
        }
        __threadfence();
        __syncthreads();
        for (int b = t; b < nbins; b += T) {  // key[b] is now the END of bin b
          const uint32_t lo = b ? key[b - 1] : 0u, hi = key[b];
          for (uint32_t i = lo + 1; i < hi; i++) {  // insertion sort (bins hold a few elements)
            const unsigned long long x = ld_coherent((const uint64_t WG_G*)out + i);
            uint32_t q = i;
            while (q > lo && ld_coherent((const uint64_t WG_G*)out + (q - 1)) > x) {
              out[q] = ld_coherent((const uint64_t WG_G*)out + (q - 1));
              q--;
            }
            out[q] = x;
          }
        }
        __threadfence();
        __syncthreads();
        uint32_t mineB = 0;
        for (int pos = t; pos < m; pos += T) {
          const unsigned long long kv = ld_coherent((const uint64_t WG_G*)out + pos);
          h_peer_store(s, (size_t)snd * (N - 1) + (m - 1) + pos, base + (int32_t)(uint32_t)kv, (uint32_t)(kv >> 32));
          mineB += pos == 0 || (uint32_t)(kv >> 32) != (uint32_t)(ld_coherent((const uint64_t WG_G*)out + (pos - 1)) >> 32);
        }
        mineB = wave_reduce_add32(mineB);
        if (WG_LANE == 0 && mineB) atomicAdd(&shGroups, mineB);
        __syncthreads();
        if (t == 0) cnt[(size_t)snd * L + l] = (uint32_t)m - shGroups;
        __syncthreads();
        continue;
      }
      for (int k = threadIdx.x; k < m; k += blockDim.x)
        key[k] = ((uint32_t)s.ranks[(size_t)(base + k) * N + snd] << idBits) | (uint32_t)k;
      if (threadIdx.x == 0) shGroups = 0;
      __syncthreads();
      for (int kk = 2; kk <= m; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
          for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const int ixj = i ^ j;
            if (ixj > i) {
              const uint32_t a = key[i], b = key[ixj];
              if ((a > b) == ((i & kk) == 0)) {
                key[i] = b;
                key[ixj] = a;
              }
            }
          }
          __syncthreads();
        }
      uint32_t mine = 0;
      for (int pos = threadIdx.x; pos < m; pos += blockDim.x) {
        const uint32_t kv = key[pos];
        h_peer_store(s, (size_t)snd * (N - 1) + (m - 1) + pos, base + (int32_t)(kv & (uint32_t)(m - 1)), kv >> idBits);
        mine += pos == 0 || (kv >> idBits) != (key[pos - 1] >> idBits);  // a bucket starts here
      }
      mine = wave_reduce_add32(mine);
      if (WG_LANE == 0 && mine) atomicAdd(&shGroups, mine);
      __syncthreads();
      if (threadIdx.x == 0) cnt[(size_t)snd * L + l] = (uint32_t)m - shGroups;  // draws: a bucket of g draws g - 1 times
      __syncthreads();
    }
  }
}
// (B): one thread per list position; the first position of a bucket of g >= 2 receivers runs the bucket's shuffle —
// for (k = g; k > 1; k--) swap(a[k - 1], a[rd.nextInt(k)]) (Collections.shuffle) — from the jumped rd state
__global__ void __launch_bounds__(256) k_handel_init_shuffle(HandelState s, const uint8_t* __restrict__ down,
                                                             const unsigned long long* __restrict__ offs, uint64_t rng0,
                                                             uint32_t* __restrict__ rejected) {
  __shared__ uint32_t shW[4];
  __shared__ uint32_t shCarry;
  const int N = s.N, L = s.L;
  for (int snd = blockIdx.x; snd < N; snd += gridDim.x) {
    if (down[snd]) continue;
    for (int l = 2; l < L; l++) {  // (level 1 is one receiver)
      const int m = 1 << (l - 1);
      const size_t at = (size_t)snd * (N - 1) + (m - 1);
      const unsigned long long off = offs[(size_t)snd * L + l];
      if (threadIdx.x == 0) shCarry = 0;
      __syncthreads();
      for (int p0 = 0; p0 < m; p0 += blockDim.x) {
        const int pos = p0 + (int)threadIdx.x;
        int32_t rk = -1, rkPrev = -2, rkNext = -3;
        if (pos < m) {
          rk = s.ranks[(size_t)h_peer(s, at + pos) * N + snd];
          if (pos > 0) rkPrev = s.ranks[(size_t)h_peer(s, at + pos - 1) * N + snd];
          if (pos + 1 < m) rkNext = s.ranks[(size_t)h_peer(s, at + pos + 1) * N + snd];
        }
        const bool start = pos < m && rk != rkPrev;
        // buckets started before this position (this one included): block-wide inclusive count
        const uint64_t bm = __ballot(start);
        const uint32_t inWave = (uint32_t)__popcll(bm & (lanes_lt() | (1ULL << WG_LANE)));
        if (WG_LANE == 0) shW[threadIdx.x >> 6] = (uint32_t)__popcll(bm);
        __syncthreads();
        uint32_t before = shCarry, total = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
          if (w < (int)(threadIdx.x >> 6)) before += shW[w];
          total += shW[w];
        }
        const uint32_t q = before + inWave - 1;  // this position's bucket, counted from 0
        __syncthreads();
        if (threadIdx.x == 0) shCarry += total;
        if (start && rk == rkNext) {  // a bucket of g >= 2: this is its first position
          int g = 2;
          while (pos + g < m && s.ranks[(size_t)h_peer(s, at + pos + g) * N + snd] == rk) g++;
          // draws before the bucket in this list: (positions before) - (buckets before)
          uint64_t st = lcg_skip(rng0, off + (unsigned long long)(pos - (int)q));
          for (int k = g; k > 1; k--) {
            int consumed;
            const int32_t j = lcg_next_int_bounded(st, k, &consumed);
            if (consumed != 1) atomicOr(rejected, 1u);
            const int32_t a = h_peer(s, at + pos + k - 1), b = h_peer(s, at + pos + j);  // (one bucket: the same rank)
            h_peer_store(s, at + pos + k - 1, b, (uint32_t)rk);
            h_peer_store(s, at + pos + j, a, (uint32_t)rk);
          }
        }
        __syncthreads();
      }
    }
  }
}

// ---- sharded engine: the periodic-task snapshots of this ms (Handel SendSigs.sigs, P/Handel.java:254; GSFSignature
// toSend.clone(), P/GSFSignature.java:146) reach the other shards -------------------------------------------------
// A snapshot is read at delivery by the receiver's shard at the address the message carries, so every shard keeps
// the whole snapshot ring and the rows written in this ms are summed across shards (zeros from non-owners). The
// rows are those of the periodic-task events (task word TASK) of the (replicated) event list, numbered in event
// order by SnapF. S = HandelState / GsfState: snap, snapNb, snapStride, N, snapIdx, nSnap, xsnap, xsnapRows.
__device__ __forceinline__ int32_t snap_period(const HandelState& s) { return s.p.disseminationPeriodMs; }
// 64-bit words of the snapshot row event e wrote (0: none): its owner reported the width with the event's result
// (EV_SNAP_*, exchange 1), so every shard numbers the same rows
template <uint32_t TASK, class S>
__device__ __forceinline__ uint32_t snapshot_words(const EngineDev& d, const S& s, uint32_t e) {
  const uint32_t code = (d.evRes[e].nrec & EV_SNAP_MASK) >> EV_SNAP_SHIFT;
  if (!code) return 0u;
  const Rec r = d.ev[e];
  if (rec_kind(r) != K_PERIODIC || r.w2 != TASK) return 0u;
  return min(1u << (code - 1), s.snapStride);
}
template <class S, uint32_t TASK>
struct SnapF {
  typedef S Aux;
  const EngineDev& d;
  const S& s;
  __device__ SnapF(const EngineDev& d_, const Aux* a) : d(d_), s(*a) {}
  __device__ uint32_t count() const { return d.g->nEvents; }
  // rows are packed back to back: the scan runs over their widths, `nSnap` is the image's size in 64-bit words
  __device__ uint64_t value(uint32_t e) const { return snapshot_words<TASK>(d, s, e); }
  __device__ void tally(uint32_t, uint32_t) const {}
  __device__ void total(uint64_t tot) const {
    const uint64_t cap = (uint64_t)s.xsnapRows * s.snapStride;
    if (tot > cap) set_err(d.g, ERR_PAYLOAD);
    *s.nSnap = (uint32_t)(tot > cap ? cap : tot);
  }
  __device__ void write(uint32_t e, uint64_t excl, bool valid) const {
    if (valid) s.snapIdx[e] = (uint32_t)excl;  // the row's first word in the image
  }
};
// one wavefront per snapshot event: pack = owner's ring row -> image (zeros elsewhere); unpack = summed
// image -> ring row on the shards that do not own the node
template <class S, uint32_t TASK, bool PACK>
__global__ void __launch_bounds__(256) k_shard_snap(const EngineDev* __restrict__ tab, const S* __restrict__ stab) {
  WG_ENGINE(tab);
  const S& s = stab[wgBy];
  const uint32_t wave = (wgBx * blockDim.x + threadIdx.x) >> 6, nWaves = (wgGx * blockDim.x) >> 6;
  const uint32_t nEv = d.g->nEvents;
  const uint32_t win = ((uint32_t)d.g->now / (uint32_t)snap_period(s)) % s.snapNb;
  const uint64_t cap = (uint64_t)s.xsnapRows * s.snapStride;
  for (uint32_t e = wave; e < nEv; e += nWaves) {
    const uint32_t w = snapshot_words<TASK>(d, s, e);
    if (!w || (uint64_t)s.snapIdx[e] + w > cap) continue;
    const int32_t node = (int32_t)d.ev[e].w1;
    const bool owned = shard_owns(d, node);
    uint64_t WG_G* row = s.snap + (size_t)(win * (uint32_t)s.N + (uint32_t)node) * s.snapStride;
    uint64_t WG_G* img = (uint64_t WG_G*)(int32_t WG_G*)s.xsnap + (size_t)s.snapIdx[e];
    for (uint32_t j = WG_LANE; j < w; j += 64) {
      if (PACK)
        img[j] = owned ? row[j] : 0ULL;
      else if (!owned)
        row[j] = img[j];
    }
  }
}

// ---- owner-directed snapshot exchange (HandelState::xout / xin): this shard's row of the count matrix, and the chunks
// the other shards sent copied into this shard's ring at the offset each one names
// (a region that overflowed — ERR_PAYLOAD, reported at the end of the ms — publishes its capacity, not the count the atomic ran
// up to: the all-to-all must never be sized past the region, let alone past the allocation behind the last one)
__global__ void k_handel_xcounts(HandelState s) {
  const int d = (int)threadIdx.x;
  if (d < s.xS) s.xcounts[s.xMe * s.xS + d] = (int32_t)min(s.xoutCount[d], s.xChunkCap);
}
__global__ void __launch_bounds__(256) k_handel_xunpack(HandelState s, uint32_t nChunks) {
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, j = threadIdx.x & 15u, stride = (gridDim.x * blockDim.x) >> 4;
  for (uint32_t i = g; i < nChunks; i += stride) {
    const uint64_t WG_G* c = s.xin + (size_t)i * H_XCHUNK;
    const uint64_t h = c[0];
    if (j < (uint32_t)(h >> 32)) s.snap[(uint32_t)h + j] = c[1 + j];
  }
}

}  // namespace wg
