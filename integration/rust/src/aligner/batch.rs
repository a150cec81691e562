//! Additive batch entry points of `libparasail_amd.so` for parasail-rs' `Aligner`.
//!
//! NOT COMPILED in the image this repository is built in (no rustc / cargo there): written against parasail-rs as found under
//! `/root/reference` (`src/aligner/mod.rs`, `src/alignment/mod.rs`, `src/matrix/mod.rs`, `src/profile/mod.rs`) and against the
//! C declarations in `include/parasail_amd.h`, which are exercised through the ctypes and C++ mirrors of this same interface.
//!
//! `Aligner::align()` keeps the reference's one-pair semantics (`src/aligner/mod.rs:397-452`); the functions here hand MANY
//! independent pairs to the GPU in one call, which is where the throughput is (one pair cannot fill 256 compute units).
//!
//! Fields to add to `struct Aligner` (`src/aligner/mod.rs:372-382`), filled in `AlignerBuilder::build()` (`:339-369`) from what
//! the builder already knows:
//!
//! ```ignore
//! // batch: 0 nw, 1 sg, 2 sw                    <- self.mode  ("nw" | "sg" | "sw", :59-61)
//! pub(crate) mode_id: i32,
//! // batch: 1 query begin | 2 query end | 4 ref begin | 8 ref end are free  <- allow_query_gaps / allow_ref_gaps (:270-299);
//! //        plain "sg" = 15
//! pub(crate) sg_flags: i32,
//! // batch: 0 = sat, 8, 16, 32, 64               <- self.solution_width (:125-137)
//! pub(crate) width: i32,
//! // batch: use_stats / use_trace                <- self.use_stats, self.use_trace (:210-267)
//! pub(crate) want_stats: bool,
//! pub(crate) want_trace: bool,
//! ```
use crate::{Aligner, Error, Result};
use libparasail_sys::{parasail_matrix_t, parasail_profile_t};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

/// `pmx_config_t` (include/parasail_amd.h)
#[repr(C)]
pub struct PmxConfig {
    pub mode: c_int,     // 0 nw, 1 sg, 2 sw
    pub sg_flags: c_int, // 1 qb | 2 qe | 4 db | 8 de
    pub open: c_int,
    pub extend: c_int,
    pub width: c_int, // 0 sat, 8, 16, 32, 64
    pub want: c_int,  // 1 stats | 2 cigar | 4 ragged lengths: process in length-sorted order
    pub matrix: *const parasail_matrix_t,
}

/// `pmx_record_t`: what `Alignment::{get_score,get_end_query,get_end_ref}` return (`src/alignment/mod.rs:64-76`), plus
/// `flags` (bit 0: saturated in the requested width -- `Alignment::is_saturated`, `:436-440`).
#[repr(C)]
#[derive(Clone, Copy, Default, Debug, PartialEq, Eq)]
pub struct PmxRecord {
    pub score: i32,
    pub end_query: i32,
    pub end_ref: i32,
    pub flags: i32,
}

/// `pmx_stats_t`: `Alignment::{get_matches,get_similar,get_length}` (`src/alignment/mod.rs:79-98`).
#[repr(C)]
#[derive(Clone, Copy, Default, Debug, PartialEq, Eq)]
pub struct PmxStats {
    pub matches: i32,
    pub similar: i32,
    pub length: i32,
}

pub const PMX_WANT_STATS: c_int = 1;
pub const PMX_WANT_CIGAR: c_int = 2;
pub const PMX_WANT_SORTED: c_int = 4;

#[link(name = "parasail_amd")]
extern "C" {
    fn pmx_align_batch(
        cfg: *const PmxConfig, n: i64,
        qbuf: *const u8, qoff: *const i64, rbuf: *const u8, roff: *const i64,
        out: *mut PmxRecord, stats_out: *mut PmxStats,
    ) -> c_int;
    fn pmx_align_profile_batch(
        cfg: *const PmxConfig, profile: *const parasail_profile_t, n: i64,
        rbuf: *const u8, roff: *const i64,
        out: *mut PmxRecord, stats_out: *mut PmxStats,
    ) -> c_int;
    fn pmx_align_batch_cigar(
        cfg: *const PmxConfig, n: i64,
        qbuf: *const u8, qoff: *const i64, rbuf: *const u8, roff: *const i64,
        out: *mut PmxRecord, cigar_buf: *mut *mut c_char, cigar_off: *mut i64,
    ) -> c_int;
    fn pmx_align_batch_banded(
        cfg: *const PmxConfig, profile: *const parasail_profile_t, n: i64,
        qbuf: *const u8, qoff: *const i64, rbuf: *const u8, roff: *const i64,
        band: i32, diag: *const i32, out: *mut PmxRecord,
    ) -> c_int;
    fn pmx_align_batch_multi(
        cfg: *const PmxConfig, n: i64,
        qbuf: *const u8, qoff: *const i64, rbuf: *const u8, roff: *const i64,
        devices: *const c_int, ndev: c_int, out: *mut PmxRecord, stats_out: *mut PmxStats,
    ) -> c_int;
    fn pmx_align_profile_batch_multi(
        cfg: *const PmxConfig, profile: *const parasail_profile_t, n: i64,
        rbuf: *const u8, roff: *const i64,
        devices: *const c_int, ndev: c_int, out: *mut PmxRecord, stats_out: *mut PmxStats,
    ) -> c_int;
    fn pmx_align_batch_2bit(
        cfg: *const PmxConfig, n: i64,
        q2: *const u8, qoff: *const i64, r2: *const u8, roff: *const i64,
        out: *mut PmxRecord, stats_out: *mut PmxStats,
    ) -> c_int;
    // device-pointer entries (asynchronous on a hipStream_t): for callers that already hold their sequences in HBM
    pub fn pmx_align_batch_device(
        cfg: *const PmxConfig, n: i64,
        d_qbuf: *const u8, d_qoff: *const i64, d_rbuf: *const u8, d_roff: *const i64,
        max_qlen: i32, max_rlen: i32,
        d_out: *mut PmxRecord, d_stats: *mut PmxStats, stream: *mut c_void,
    ) -> c_int;
    pub fn pmx_align_profile_batch_device(
        cfg: *const PmxConfig, profile: *const parasail_profile_t, n: i64,
        d_rbuf: *const u8, d_roff: *const i64, max_rlen: i32,
        d_out: *mut PmxRecord, d_stats: *mut PmxStats, stream: *mut c_void,
    ) -> c_int;
    pub fn pmx_align_batch_cigar_device(
        cfg: *const PmxConfig, n: i64,
        d_qbuf: *const u8, d_qoff: *const i64, d_rbuf: *const u8, d_roff: *const i64,
        max_qlen: i32, max_rlen: i32, d_out: *mut PmxRecord,
        d_cigar_text: *mut c_char, cigar_capacity: i64, d_cigar_off: *mut i64, stream: *mut c_void,
    ) -> c_int;
    pub fn pmx_align_batch_table_device(
        cfg: *const PmxConfig, n: i64,
        d_qbuf: *const u8, d_qoff: *const i64, d_rbuf: *const u8, d_roff: *const i64,
        max_qlen: i32, max_rlen: i32, d_tab_off: *const i64, d_score_table: *mut i32,
        d_score_row: *mut i32, d_score_col: *mut i32, d_out: *mut PmxRecord, stream: *mut c_void,
    ) -> c_int;
    pub fn pmx_shard_bounds_by_cells(n: i64, qoff: *const i64, roff: *const i64, parts: c_int, bounds: *mut i64) -> c_int;
    pub fn pmx_host_register(p: *mut c_void, bytes: usize) -> c_int;
    pub fn pmx_host_unregister(p: *mut c_void) -> c_int;
    fn pmx_free(p: *mut c_void);
    fn pmx_last_error() -> *const c_char;
    pub fn pmx_last_kernel() -> *const c_char;
    pub fn pmx_switches() -> *const c_char;
}

/// Sequences packed back to back with `n + 1` byte offsets: the layout of every `pmx_*` batch entry.
pub struct Packed {
    pub buf: Vec<u8>,
    pub off: Vec<i64>,
}

impl Packed {
    pub fn from_slices(seqs: &[&[u8]]) -> Packed {
        let mut off = Vec::with_capacity(seqs.len() + 1);
        let mut buf = Vec::with_capacity(seqs.iter().map(|s| s.len()).sum());
        off.push(0i64);
        for s in seqs {
            buf.extend_from_slice(s);
            off.push(buf.len() as i64);
        }
        Packed { buf, off }
    }
    pub fn len(&self) -> usize {
        self.off.len() - 1
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
}

/// Records, optional statistics.
pub struct BatchResult {
    pub records: Vec<PmxRecord>,
    pub stats: Option<Vec<PmxStats>>,
}

/// CIGAR text of a batch: one callee-allocated block, released with `pmx_free` on drop
/// (the per-pair `CigarString` of `src/alignment/mod.rs:32-44` owns its block the same way).
pub struct BatchCigars {
    text: *mut c_char,
    off: Vec<i64>,
}

impl BatchCigars {
    pub fn len(&self) -> usize {
        self.off.len() - 1
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
    /// CIGAR text of pair `k` (e.g. `"93=1X20=2D134="`), the format `Alignment::get_cigar` returns (`src/alignment/mod.rs:390-419`).
    pub fn get(&self, k: usize) -> &str {
        let (a, e) = (self.off[k] as usize, self.off[k + 1] as usize);
        // digits and the letters = X I D only
        unsafe { std::str::from_utf8_unchecked(std::slice::from_raw_parts(self.text.add(a) as *const u8, e - a)) }
    }
}

impl Drop for BatchCigars {
    fn drop(&mut self) {
        if !self.text.is_null() {
            unsafe { pmx_free(self.text as *mut c_void) }
        }
    }
}

unsafe impl Send for BatchCigars {}

fn last_error() -> Error {
    Error::Batch(unsafe { CStr::from_ptr(pmx_last_error()) }.to_string_lossy().into_owned())
}

impl Aligner {
    fn pmx_config(&self, want: c_int) -> PmxConfig {
        PmxConfig {
            mode: self.mode_id,
            sg_flags: self.sg_flags,
            open: self.gap_open,
            extend: self.gap_extend,
            width: self.width,
            want,
            matrix: **self.matrix,
        }
    }

    /// Align many independent pairs in one call.  With a profile (`AlignerBuilder::profile`) pass `None` for the queries, as
    /// `align()` does (`src/aligner/mod.rs:394-396`): every reference is aligned against the profile's query.
    pub fn align_batch(&self, queries: Option<&Packed>, references: &Packed) -> Result<BatchResult> {
        let n = references.len();
        let mut records = vec![PmxRecord::default(); n];
        let with_profile = !self.profile.is_null();
        let want_stats = if with_profile { self.profile.use_stats } else { self.want_stats };
        let mut stats = if want_stats { Some(vec![PmxStats::default(); n]) } else { None };
        let stats_ptr = stats.as_mut().map_or(std::ptr::null_mut(), |s| s.as_mut_ptr());
        let cfg = self.pmx_config(if want_stats { PMX_WANT_STATS } else { 0 });
        let rc = if with_profile {
            unsafe {
                pmx_align_profile_batch(&cfg, **self.profile, n as i64, references.buf.as_ptr(), references.off.as_ptr(),
                                        records.as_mut_ptr(), stats_ptr)
            }
        } else {
            let q = queries.expect("Query sequences are required for alignment without a profile.");
            assert_eq!(q.len(), n, "one query per reference");
            unsafe {
                pmx_align_batch(&cfg, n as i64, q.buf.as_ptr(), q.off.as_ptr(), references.buf.as_ptr(), references.off.as_ptr(),
                                records.as_mut_ptr(), stats_ptr)
            }
        };
        if rc != 0 {
            return Err(last_error());
        }
        Ok(BatchResult { records, stats })
    }

    /// Score, end positions and CIGAR text per pair (an aligner built with `use_trace()`).
    pub fn align_batch_cigar(&self, queries: &Packed, references: &Packed) -> Result<(Vec<PmxRecord>, BatchCigars)> {
        let n = references.len();
        assert_eq!(queries.len(), n, "one query per reference");
        let mut records = vec![PmxRecord::default(); n];
        let mut off = vec![0i64; n + 1];
        let mut text: *mut c_char = std::ptr::null_mut();
        let cfg = self.pmx_config(PMX_WANT_CIGAR);
        let rc = unsafe {
            pmx_align_batch_cigar(&cfg, n as i64, queries.buf.as_ptr(), queries.off.as_ptr(),
                                  references.buf.as_ptr(), references.off.as_ptr(),
                                  records.as_mut_ptr(), &mut text, off.as_mut_ptr())
        };
        if rc != 0 {
            return Err(last_error());
        }
        Ok((records, BatchCigars { text, off }))
    }

    /// Banded batch: the extension of `banded_nw` (`src/aligner/mod.rs:454-489`) to many pairs, any mode, and an optional
    /// per-pair band centre (cells with `|(j - i) - diag[k]| > bandwidth` are excluded).
    pub fn banded_batch(&self, queries: Option<&Packed>, references: &Packed, diag: Option<&[i32]>) -> Result<Vec<PmxRecord>> {
        let band = self.bandwidth.ok_or(Error::NoBandwidth)?;
        let n = references.len();
        let mut records = vec![PmxRecord::default(); n];
        let cfg = self.pmx_config(0);
        let (qb, qo) = queries.map_or((std::ptr::null(), std::ptr::null()), |q| (q.buf.as_ptr(), q.off.as_ptr()));
        let prof = if self.profile.is_null() { std::ptr::null() } else { **self.profile as *const parasail_profile_t };
        let rc = unsafe {
            pmx_align_batch_banded(&cfg, prof, n as i64, qb, qo, references.buf.as_ptr(), references.off.as_ptr(),
                                   band, diag.map_or(std::ptr::null(), |d| d.as_ptr()), records.as_mut_ptr())
        };
        if rc != 0 {
            return Err(last_error());
        }
        Ok(records)
    }

    /// One process driving several GPUs of the node: contiguous blocks of about equal cell counts, one per listed device;
    /// records come back in input order.
    pub fn align_batch_multi(&self, queries: Option<&Packed>, references: &Packed, devices: &[i32]) -> Result<BatchResult> {
        let n = references.len();
        let mut records = vec![PmxRecord::default(); n];
        let with_profile = !self.profile.is_null();
        let want_stats = if with_profile { self.profile.use_stats } else { self.want_stats };
        let mut stats = if want_stats { Some(vec![PmxStats::default(); n]) } else { None };
        let stats_ptr = stats.as_mut().map_or(std::ptr::null_mut(), |s| s.as_mut_ptr());
        let cfg = self.pmx_config(if want_stats { PMX_WANT_STATS } else { 0 });
        let rc = if with_profile {
            unsafe {
                pmx_align_profile_batch_multi(&cfg, **self.profile, n as i64, references.buf.as_ptr(), references.off.as_ptr(),
                                              devices.as_ptr(), devices.len() as c_int, records.as_mut_ptr(), stats_ptr)
            }
        } else {
            let q = queries.expect("Query sequences are required for alignment without a profile.");
            unsafe {
                pmx_align_batch_multi(&cfg, n as i64, q.buf.as_ptr(), q.off.as_ptr(), references.buf.as_ptr(), references.off.as_ptr(),
                                      devices.as_ptr(), devices.len() as c_int, records.as_mut_ptr(), stats_ptr)
            }
        };
        if rc != 0 {
            return Err(last_error());
        }
        Ok(BatchResult { records, stats })
    }

    /// 2-bit packed DNA (base b in byte b / 4 at bits 2 * (b % 4); code c = letter c of the matrix alphabet; offsets count
    /// bases): a quarter of the bytes cross PCIe.
    pub fn align_batch_2bit(&self, q2: &[u8], qoff: &[i64], r2: &[u8], roff: &[i64]) -> Result<Vec<PmxRecord>> {
        let n = roff.len() - 1;
        assert_eq!(qoff.len(), roff.len());
        let mut records = vec![PmxRecord::default(); n];
        let cfg = self.pmx_config(0);
        let rc = unsafe {
            pmx_align_batch_2bit(&cfg, n as i64, q2.as_ptr(), qoff.as_ptr(), r2.as_ptr(), roff.as_ptr(),
                                 records.as_mut_ptr(), std::ptr::null_mut())
        };
        if rc != 0 {
            return Err(last_error());
        }
        Ok(records)
    }
}

#[cfg(test)]
mod tests {
    //! The reference's KAT style (`tests/test_parasail.rs:65-122`), on batches.
    use super::*;
    use crate::Matrix;

    #[test]
    fn batch_of_identical_pairs() -> crate::Result<()> {
        let q: Vec<&[u8]> = vec![b"ACGT"; 1000];
        let (qs, rs) = (Packed::from_slices(&q), Packed::from_slices(&q));
        let aligner = Aligner::new().local().build();
        let out = aligner.align_batch(Some(&qs), &rs)?;
        assert!(out.records.iter().all(|r| (r.score, r.end_query, r.end_ref) == (4, 3, 3)));
        Ok(())
    }

    #[test]
    fn batch_cigar() -> crate::Result<()> {
        let matrix = Matrix::create(b"ACGT", 2, -3)?;
        let q: Vec<&[u8]> = vec![b"ACGTACGTAC"; 64];
        let r: Vec<&[u8]> = vec![b"ACGTACGTAC"; 64];
        let aligner = Aligner::new().global().matrix(matrix).gap_open(5).gap_extend(2).use_trace().build();
        let (rec, cig) = aligner.align_batch_cigar(&Packed::from_slices(&q), &Packed::from_slices(&r))?;
        assert_eq!(rec[0].score, 20);
        assert_eq!(cig.get(0), "10=");
        Ok(())
    }
}
