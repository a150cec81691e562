/*
 * pmx_conventions.h -- the ONE place where the conventions the reference does not pin are chosen.
 *
 * parasail-rs only PRINTS its CIGAR / traceback strings in its tests (tests/test_parasail.rs:581-616), and the library that
 * produces them (libparasail-sys 0.2.1 -> jeffdaily/parasail) is not in this image, so the letters below cannot be checked
 * against the reference here.  Product (HIP kernels, C ABI) and checker (oracle/) both read them from this header: a flip is a
 * one-line change.
 *
 * Trace states (flag names pinned by src/alignment/table.rs:127-142): INS = the E table (horizontal move: consumes a character
 * of s2 / the reference, gap character in the query line), DEL = the F table (vertical move: consumes a character of s1 / the
 * query).  Chosen letters: SAM sense with query = s1, reference = s2 -- consuming only the reference is 'D', consuming only the
 * query is 'I'.  The other orientation (state INS -> 'I', state DEL -> 'D', which a round-1 review note believes upstream's
 * cigar.c uses) is available WITHOUT a rebuild: set PMX_CIGAR_SWAP_ID=1 in the environment of the calling process and every
 * CIGAR handed out (packed ops of parasail_result_get_cigar / parasail_ssw, decoded text, batch CIGAR text) has I and D
 * exchanged; the defines below are the compiled default ("CIGAR letters").  Evidence for this choice (outside the reference tree, from memory): downstream users hand (read, reference) to
 * parasail as (s1, s2) and feed the decoded CIGAR to SAM/BAM writers unchanged apart from clipping.  [UNPINNED]
 */
#ifndef PMX_CONVENTIONS_H
#define PMX_CONVENTIONS_H

#define PMX_CIGAR_LETTER_FOR_INS_STATE 'D'   /* E / horizontal / consumes a reference character */
#define PMX_CIGAR_LETTER_FOR_DEL_STATE 'I'   /* F / vertical   / consumes a query character     */

/* the same as BAM op codes ("MIDNSHP=X": I = 1, D = 2) */
#define PMX_BAM_OP_FOR_INS_STATE (PMX_CIGAR_LETTER_FOR_INS_STATE == 'D' ? 2u : 1u)
#define PMX_BAM_OP_FOR_DEL_STATE (PMX_CIGAR_LETTER_FOR_DEL_STATE == 'I' ? 1u : 2u)

#endif
