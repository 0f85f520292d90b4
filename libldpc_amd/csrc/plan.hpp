// plan.hpp — device execution plan for a parity-check graph.
//
// The reference walks H through per-node neighbour vectors (sparse.h:77-79) one frame at a time.
// The MI355X decoder keeps one frame's messages in LDS and gives every lane one node, so the graph
// is re-laid-out once per code:
//
//   * check nodes are grouped by degree into blocks of <= 64 (one wave, uniform degree, no
//     divergence); block b owns message slots [off, off + degree*count), slot(lane, j) =
//     off + j*count + lane, so a wave's j-th load/store is one contiguous, bank-conflict-free run;
//   * variable nodes are grouped the same way; VN lane `rank` reaches its edges (in the file order
//     the reference sums them in, decoder.cpp:50-56) through a u16 slot table laid out
//     [block][position][lane] so the index loads coalesce;
//   * blocks are dealt to the workgroup's waves by longest-processing-time so that waves finish
//     each half-iteration together.
//
// Message values, their summation order and the per-edge results are those of the reference;
// only where they live changes.
#pragma once

#include <cstdint>
#include <vector>

#include "code.hpp"

namespace ldpc_amd
{

constexpr int kWaveSize = 64;
constexpr int kMaxLdsCnDegree = 8;  // widest check node of the LDS-resident decoder's register CN update
constexpr int kMaxCnDegree = 16;    // widest check node of the memory-resident decoder
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr int kDecodeWaves = 4;     // waves per workgroup of the LDS-resident decoder

struct CnBlock
{
    uint32_t off;    // first message slot
    uint16_t count;  // check nodes in the block (<= 64)
    uint16_t degree;
};

constexpr int kVnPackedRows = 32;
constexpr uint32_t kVnSrcZero = 0xFFFFFFFFu, kVnSrcShortened = 0xFFFFFFFEu;

struct VnBlock
{
    uint32_t idx_off; // into vn_slot: [idx_off + p*count + lane]
    uint32_t first;   // first VN rank
    uint16_t count;
    uint16_t degree;
};

// Host-side plan (uploaded verbatim by the engine)
struct Plan
{
    int nc = 0, mc = 0, nnz = 0, nct = 0;
    int n_bitpos = 0; // entries of bit_pos (> nct when a column is both punctured and shortened)
    int max_cn_degree = 0, max_vn_degree = 0;
    std::vector<CnBlock> cn_blocks;
    std::vector<VnBlock> vn_blocks;
    std::vector<uint32_t> vn_slot;   // slot of the p-th edge (column file order) of each VN
    std::vector<uint32_t> cn_work;   // [kDecodeWaves][cn_work_stride] block ids, 0xFFFF = none
    std::vector<uint32_t> vn_work;   // [kDecodeWaves][vn_work_stride]
    int cn_work_stride = 0, vn_work_stride = 0;
    // the same work lists with the descriptors in place of the block ids (count 0 = none; cn_work_desc rows are
    // padded to an even number of entries plus two): a wave reads the two blocks it handles together with ONE
    // scalar load instead of two dependent ones
    std::vector<CnBlock> cn_work_desc; // [kDecodeWaves][cn_desc_stride]
    int cn_desc_stride = 0;
    std::vector<uint32_t> vn_work_desc; // [kDecodeWaves][vn_work_stride + 1][4]: idx_off, first, count | degree << 16, 0
    // the slot indices a lane of the LDS-resident decoder keeps in registers, as it keeps them (two u16 per word), so that it
    // picks them up with one load per word, none dependent on another: [kDecodeWaves][kVnPackedRows][64] — rows 0..7: block
    // w of the wave's work list if its nodes have one or two edges (first slot | last slot << 16), rows 8..15: the wave's
    // FIRST block if its nodes have 3..16 edges (slots 2i, 2i+1 in word i); zero where there is nothing.  Rows 16..23: where
    // the LLR of the lane's node in block w comes from — the index of its bit among the transmitted ones, or kVnSrcZero
    // (punctured, never written by the channel, no node) / kVnSrcShortened; rows 24..31: the node's column (kVnSrcZero: no
    // node).  Empty when a slot needs more than 16 bits or a wave has more than 8 blocks (such codes do not use the
    // register-held indices).
    std::vector<uint32_t> vn_packed;
    std::vector<uint32_t> col_rank;  // column -> VN rank
    std::vector<uint32_t> rank_col;  // VN rank -> column
    std::vector<uint32_t> tx_rank;   // transmitted index i -> rank of bit_pos[i]
    std::vector<uint8_t> rank_kind;  // 0 transmitted, 1 punctured, 2 shortened, 3 never written by the channel
    std::vector<uint32_t> rank_slot0; // slot of the VN's first edge, kNoSlot for an isolated VN
    std::vector<uint32_t> edge_slot; // file-order edge -> slot (tests / debugging)
    std::vector<uint32_t> cn_rank_row; // slot-space CN order -> original row (tests)
    size_t lds_bytes = 0;            // dynamic LDS the LDS-resident kernel needs per frame
    bool lds_ok = false;             // fits the LDS-resident kernel's limits
    bool hbm_ok = false;             // within the memory-resident kernel's limits
    bool has_isolated_vn = false;    // a column without any edge
};

// ---- register-resident decoder (codes too large for 4 frames of LDS per CU, e.g. n=8192) ----------------
// One workgroup of nt threads = one frame.  Thread (wave, lane) owns the check nodes of the CN blocks
// k*(nt/64) + wave, k < kc, and keeps their messages in registers m[k][j].  The variable-node side is reached through
// an LDS mailbox laid out VN-block-major ([position][lane] inside a block of <= 64 equal-degree VNs); a code
// whose edges do not fit the mailbox at once is exchanged in `rounds` groups of VN blocks.
constexpr uint32_t kRegNoEdge = 0xFFFFFFFFu;

struct RegVnBlock
{
    uint32_t first;   // first VN rank
    uint32_t mb_off;  // mailbox offset (in doubles) of (position 0, lane 0); (p, lane) sits at mb_off + p*count + lane
    uint16_t count;
    uint16_t degree;
};

struct RegPlan
{
    bool ok = false;
    int nt = 0;                     // threads per workgroup (1024: one frame per CU, 512: two)
    int kc = 0, maxd = 0;           // CN blocks per wave, register columns per CN
    int rounds = 0;
    uint32_t mb_doubles = 0;        // mailbox capacity in doubles (LDS = 9 bytes per entry)
    std::vector<uint32_t> cn_edge;  // [(k*maxd + j)*nt + tid] = (round << 28) | mailbox offset, kRegNoEdge = none
    std::vector<uint8_t> cn_deg;    // [k*(nt/64) + wave] degree of that CN block (0 = none)
    std::vector<uint8_t> cn_cnt;    // [k*(nt/64) + wave] check nodes in the block
    std::vector<RegVnBlock> vn_blocks;      // in round order
    std::vector<uint32_t> round_first;      // [rounds+1] first VN block of each round
};

// ---- register-resident decoder, second form: totals come back instead of messages ---------------------------
// One workgroup of nt threads (nt/64 waves) = one frame per CU.  Thread (wave, lane) owns the check nodes of CN blocks
// k*(nt/64) + wave, k < kc, and keeps their messages in registers.  The thread that owns an edge also forms the edge's
// v2c message: the variable node's thread returns only the node's TOTAL (decoder.cpp:50-56: out, or its likelihood
// ratio), one entry per node instead of one per edge, and the hard decision travels in that entry's sign
// (ratio form) or is read off it (LLR domain: out <= 0).  c2v messages reach the variable nodes through an LDS
// mailbox of 8-byte entries in two rounds (the n=8192 code's 24576 messages exceed 160 KB):
//
//   entries [0, e_max)            mailbox: round 0 uses [0, E0), round 1 [0, E1); a VN block's column sits at
//                                 p0_off + lane (position 0) and prest_off + (p-1)*count + lane (positions >= 1)
//   entries [e_max, e_max + n0)   totals of the round-0 variable nodes (round-1 totals are written in place, over
//                                 position 0 of their column, which nothing overwrites before the gather)
//   entry   16384                 trash, at byte 0x20000 exactly: every thread writes all its register columns in
//                                 both rounds, and the address arithmetic sends the columns whose node belongs to
//                                 the other round (or that hold no edge) here (edge_w below)
//   entry   neutral               +1.0: what register columns without an edge gather
//   (when the round-0 totals do not fit below the trash entries they follow them)
//
// Variable-node blocks are dealt to (round, i, wave): wave w handles blocks [(round, i, w)] for i < nv[round].

constexpr uint32_t kReg2TrashEntry = 16384; // byte 0x20000

struct Reg2VnBlock
{
    uint32_t p0_off;    // entry of (position 0, lane 0)
    uint32_t prest_off; // entry of (position 1, lane 0)
    uint32_t tot_off;   // entry the total of lane 0 is written to
    uint16_t count;     // 0 = no block
    uint16_t degree;
};

struct Reg2Plan
{
    bool ok = false;
    int nt = 0, kc = 0, maxd = 0, nv0 = 0, nv1 = 0;
    uint32_t e_max = 0, neutral = 0, lds_entries = 0;
    bool uniform_cn = false; // all kc * (nt/64) check-node blocks exist and have maxd edges: the kernel's UCN instantiation
    bool uniform_vn = false; // all (nv0 + nv1) * (nt/64) variable-node blocks are full, of degree 3, at affine offsets:
    uint32_t vn_affine[2][6] = {}; // per round {p0 base, stride, prest base, stride, tot base, stride} in entries per block
    // [(k*maxd + j)*nt + tid]: bits 3..17 = byte address the edge's VN total is gathered from, bits 18..31 = mailbox
    // entry the edge's c2v message is scattered to, bit 0 = the round of the edge's node, bit 1 = no edge.  Rotated
    // right by 15 and masked with 0x7FFF8 the word is the scatter byte address plus 0x20000 (round 1) or 0x40000 (no
    // edge): round r subtracts r * 0x20000 and takes the unsigned minimum with the trash address 0x20000, which leaves
    // the addresses of its own round alone and sends everything else to trash.
    std::vector<uint32_t> edge_w;
    std::vector<uint8_t> cn_deg;       // [k*(nt/64) + wave] degree of that CN block (0 = none)
    std::vector<uint8_t> cn_cnt;       // [k*(nt/64) + wave] check nodes in the block
    std::vector<Reg2VnBlock> vn_blocks; // [(i*(nt/64) + wave)], i < nv0 + nv1 (i < nv0: round 0)
    std::vector<uint32_t> vn_rank;      // [(i*(nt/64) + wave)*64 + lane] VN rank of the Plan (kNoSlot = none)
};

// ---- layered schedule of the opt-in non-parity modes (kernels_layered.hip) -----------------------------------------
// A sweep over the check nodes is a sequence of STEPS; a step is up to 64 check nodes of equal degree no two of which
// share a variable node (one wave = one frame, one lane = one check node of the step).  Edge j of lane l of a step sits
// at offset off + j * 64 + l of the vn table (VN rank of the Plan) and of the frame's message array.
struct LayerStep
{
    uint32_t off;
    uint16_t count, degree;
};
struct LayerPlan
{
    bool ok = false;
    std::vector<LayerStep> steps;
    std::vector<uint16_t> vn;
    uint32_t slots = 0; // entries of the message array (sum of 64 * degree over the steps)
    std::vector<int> step_of_row; // (tests: the step each check node is processed in)
};
LayerPlan build_layer_plan(const LdpcCode &code, const Plan &plan);

// ---- fused form of the likelihood-ratio iteration (kernels_fused.hip; detmath.h "Fused form", fused_rule.h) ----------
// First launch of sum-product with early termination for codes the rule takes (check nodes of degree 2..4 with at most one
// leaf each).  One workgroup (4 waves) per frame as in the LDS-resident decoder, but only edges that do NOT end in a leaf
// have a message slot, check-node blocks are uniform in CLASS (degree, leaf or not, how many outputs go to degree-2
// neighbours) with their inputs in the rule's order, and a wave's work is a list of CALLS of one or two blocks.
//   slot of (block, lane, k-th message input) = off + k * count + lane; all offsets below are BYTE offsets (< 2^16).
constexpr int kFusedVnSlots = 8;    // variable-node blocks per wave (general instantiation; the small one takes 4)
constexpr int kFusedLeafCalls = 2;  // calls with leaves per wave (general instantiation; the small one takes 1)
constexpr int kFusedLaneRows = 44;  // rows of the lane table
struct FusedCall
{
    uint32_t offs; // block 0 | block 1 << 16
    uint32_t cnts; // nodes in block 0 | nodes in block 1 << 16 (0: a call of one block; both zero: end of list)
    uint32_t cls;  // dm_fused_class(degree, flip, leaf)
    uint32_t pad;
};
struct FusedPlan
{
    bool ok = false;
    int n_slots = 0;     // message slots
    int vnb = 0, cnl = 0; // largest number of variable-node blocks / of leaf calls any wave holds
    bool has_shortened = false;
    bool need_lambda = false; // a transmitted or shortened column of degree >= 3: its node keeps lambda_ch (detmath.h "Fused form")
    // a wave that serves a block through register-held offsets (slot 0, degree 3..15) serves no other, and no block goes
    // through the slot table: what the small instantiation of the kernel is compiled for
    bool wide_exclusive = false;
    std::vector<FusedCall> leaf_calls; // [kDecodeWaves][kFusedLeafCalls]
    std::vector<FusedCall> calls;      // [kDecodeWaves][calls_stride], each row ends in a zero entry
    int calls_stride = 0;
    // [kDecodeWaves][kFusedVnSlots][4]: nodes | degree << 16, offset into vn_slot, 0, 0 (nodes 0 = none).  Slot 0 holds the
    // wave's widest block; blocks of degree 2 sit in consecutive slots from an even one on (lock-step pairs).
    std::vector<uint32_t> vn_desc;
    // per wave, four bits per slot: what the variable-node pass does there (one scalar word instead of a handful of
    // loop-invariant conditions per slot that the compiler would keep in scalar registers for the whole decode)
    uint32_t vn_prog[kDecodeWaves] = {0, 0, 0, 0};
    std::vector<uint32_t> vn_slot;     // table path (degree >= 3 outside slot 0, or wider than 15): [off + p * count + lane]
    // [kDecodeWaves][kFusedLaneRows][64], what a lane keeps or needs once per frame:
    //   rows  0..7   slot w, degree 2: slot of edge 0 | slot of edge 1 << 16
    //   rows  8..15  slot 0, degree 3..15: slots of edges 2i, 2i+1 in word i
    //   rows 16..23  slot w: stage entry of the node's channel value (index among the transmitted bits; nct = shortened,
    //                nct + 1 = punctured / never written / no node), as a byte offset (x 16)
    //   rows 24..31  slot w: the node's column | kFusedCounted (its bit is one the error count visits); kFusedNone = no node
    //   rows 32..35  leaf call c, block h (row 32 + 2c + h): stage entry of the leaf's channel value
    //   rows 36..39  ... and the leaf's column | flags
    //   rows 40..43  ... and the message slot of the leaf's edge in the GENERAL plan (Plan::edge_slot: the hand-over)
    std::vector<uint32_t> lane_tab;
    std::vector<uint32_t> edge_slot;   // file-order edge -> slot, kNoSlot for an edge that ends in a leaf (tests)
    std::vector<uint32_t> ho_map;      // [n_slots] slot -> the same edge's slot in the general plan (the hand-over hands the
                                       // frame's c2v messages to the LLR-domain kernel in that plan's order)
};
constexpr uint32_t kFusedCounted = 0x80000000u, kFusedNone = 0xFFFFFFFFu;
enum : uint32_t { kFusedVnNone = 0, kFusedVnPair = 1 /* degree 2, 64 nodes */, kFusedVn2 = 2 /* degree 2 */, kFusedVnWide = 3 /* slot 0, degree 3..15 */, kFusedVnTable = 4 };
FusedPlan build_fused_plan(const LdpcCode &code, const Plan &plan);

Plan build_plan(const LdpcCode &code);
// plan for the decode_reg2_kernel<nt, kc, maxd, nv0, nv1> instantiation; ok = false when the code does not fit it
Reg2Plan build_reg2_plan(const LdpcCode &code, const Plan &plan, int nt, int kc, int maxd, int nv0, int nv1);
// `nt` threads, `kc` x `maxd` register tile = the kernel instantiation to plan for; lds_budget = bytes of LDS
// one workgroup may use for its mailbox
RegPlan build_reg_plan(const LdpcCode &code, const Plan &plan, int nt, int kc, int maxd, uint32_t lds_budget);

} // namespace ldpc_amd
