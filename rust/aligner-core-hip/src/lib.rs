//! MI355X-backed aligners behind the reference's own trait: `HipGlobalAligner`, `HipLocalAligner` (both
//! `AlignerTrait<T, Alignment<T>>`, replacing `simple::{SimpleGlobalAligner, SimpleLocalAligner}`, simple/mod.rs:19-265) and
//! `HipPWMAligner` (`AlignerTrait<T, PWMAlignment<T>>`, replacing `pwm::PWMAligner`, pwm/mod.rs:9-126), plus the batch call
//! that replaces the ten-thread loop of `calculate_p_value` (statistics/mod.rs:255-286).
//! Goes into the reference as `aligner-core/src/hip/mod.rs`.  NOT compiled in the build image (no Rust toolchain there): the C ABI
//! it binds is exercised by tests/abi_harness.c and by the Python binding instead.
use crate::alignment::{Alignment, PWMAlignment};
use crate::alignment_result::AlignmentResult;
use crate::enums::Direction;
use crate::{AlignerTrait, BioData, Error, Heuristics, Result};
use ndarray::Array2;
use std::marker::PhantomData;
use std::os::raw::{c_char, c_int};

// ---------------------------------------------------------------- include/aligner_hip.h
#[repr(C)]
pub struct AlnParams {
    pub semantics: i32,
    pub heuristics_present: i32,
    pub del: f64,
    pub ext: f64,
    pub matrix: *const f64,
    pub rows: u32,
    pub cols: u32,
    pub row_stride: i64,
    pub outputs: u32,
    pub blank_code: u8,
    pub force_f64: u8,
    pub force_serial: u8,
    pub force_generic: u8,
    pub max_passes: u32,
} // 64 bytes
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct AlnPairResult {
    pub f: f64,
    pub score: f64,
    pub end_y: u32,
    pub end_x: u32,
    pub start_y: u32,
    pub start_x: u32,
    pub aln_len: u32,
    pub status: i32,
    pub passes: u32,
    pub flags: u32,
} // 48 bytes
#[repr(C)]
pub struct AlnCtx {
    _p: [u8; 0],
}

extern "C" {
    fn aln_create(device_id: c_int, status: *mut c_int) -> *mut AlnCtx;
    fn aln_create_multi(n_devices: c_int, device_ids: *const c_int, status: *mut c_int) -> *mut AlnCtx;
    fn aln_last_error() -> *const c_char;
    fn aln_align_pair(ctx: *mut AlnCtx, p: *const AlnParams, q: *const u8, n: usize, t: *const u8, m: usize, out: *mut AlnPairResult,
                      q_aln: *mut u8, t_aln: *mut u8, directions: *mut u8, h_matrix: *mut f64) -> c_int;
    fn aln_align_batch(ctx: *mut AlnCtx, p: *const AlnParams, seqs: *const u8, q_off: *const u64, q_len: *const u64, t_off: *const u64,
                       t_len: *const u64, n_pairs: usize, results: *mut AlnPairResult, tb_buf: *mut u8, tb_off: *const u64) -> c_int;
}

pub const CORE_GLOBAL: i32 = 0;
pub const CORE_LOCAL: i32 = 1;
pub const PWM_LOCAL: i32 = 4;
pub const OUT_SCORE: u32 = 1;
pub const OUT_TRACEBACK: u32 = 2;
pub const OUT_DIRECTIONS: u32 = 4;
pub const OUT_H: u32 = 8;

/// `AlignmentResult.{alignment_matrix, direction_matrix}` are filled only on request: nothing in the reference reads them after
/// construction, and they cost 9 bytes per cell over PCIe.  `false` (the default) leaves both 0 x 0 and runs the fast
/// kernels: 0.31 ms for a 1k x 1k pair against 1.2 ms with both matrices.
pub static WANT_MATRICES: std::sync::atomic::AtomicBool = std::sync::atomic::AtomicBool::new(false);

/// One context per process.  `ALIGNER_HIP_DEVICES=all` (or a multi-GPU host process): every visible GPU behind one context --
/// single calls go to the GPUs in turn, a batch call is sharded over them (`aln_create_multi(0, NULL)`); otherwise the GPU named by
/// `LOCAL_RANK` (one process per GPU, as `torch.distributed.run` / `mpirun` launch them), default 0.
pub fn ctx() -> *mut AlnCtx {
    use std::sync::OnceLock;
    static CTX: OnceLock<usize> = OnceLock::new();
    *CTX.get_or_init(|| {
        let mut st = 0;
        let all = std::env::var("ALIGNER_HIP_DEVICES").map(|v| v == "all").unwrap_or(false);
        let c = if all {
            unsafe { aln_create_multi(0, std::ptr::null(), &mut st) }
        } else {
            let dev = std::env::var("LOCAL_RANK").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            unsafe { aln_create(dev, &mut st) }
        };
        assert!(!c.is_null(), "aln_create failed ({st}): no MI355X / HIP runtime -- there is no CPU fallback");
        c as usize
    }) as *mut AlnCtx
}

fn params(semantics: i32, del: f64, ext: f64, mat: &Array2<f64>, heuristics: bool, outputs: u32, blank: u8) -> AlnParams {
    AlnParams {
        semantics, heuristics_present: heuristics as i32, del, ext,
        matrix: mat.as_ptr(), rows: mat.nrows() as u32, cols: mat.ncols() as u32, row_stride: mat.ncols() as i64,
        outputs, blank_code: blank, force_f64: 0, force_serial: 0, force_generic: 0, max_passes: 0,
    }
}

/// status -> what the reference does in that situation (lib.rs:47-59; the panics of simple/mod.rs:85,103,214)
fn check(st: c_int) -> Result<()> {
    match st {
        0 => Ok(()),
        1 => Err(Error::UnnecessaryArgument),                              // simple/mod.rs:49-51, pwm/mod.rs:36-38
        9 => Err(Error::MatrixShapeError),                                 // pwm/mod.rs:40-42
        2 => panic!("called `Option::unwrap()` on a `None` value"),        // simple/mod.rs:103
        3 => panic!("ndarray: index out of bounds"),                       // simple/mod.rs:85
        4 => panic!("attempt to subtract with overflow"),                  // simple/mod.rs:214
        _ => panic!("aligner_hip: status {st}: {:?}", unsafe { std::ffi::CStr::from_ptr(aln_last_error()) }),
    }
}

fn directions(dirs: Vec<u8>, shape: (usize, usize)) -> Array2<Direction> {
    Array2::from_shape_vec(shape, dirs.into_iter().map(|d| match d {
        0 => Direction::Top, 1 => Direction::Left, 2 => Direction::Diagonal, _ => Direction::Beginning }).collect()).unwrap()
}

macro_rules! hip_aligner { ($name:ident, $sem:expr, $global:expr) => {
pub struct $name<T: BioData + Into<usize> + Copy + Eq> { pub query: Vec<T>, pub target: Vec<T> }

impl<T: BioData + Into<usize> + From<usize> + Copy + Eq> AlignerTrait<T, Alignment<T>> for $name<T> {
    fn from_str_seqs(query: &str, target: &str) -> Result<Self> {
        Ok($name { query: T::str_to_vec(query)?, target: T::str_to_vec(target)? })
    }
    fn from_seqs(query: &[T], target: &[T]) -> Result<Self> {
        Ok($name { query: Vec::from(query), target: Vec::from(target) })
    }
    fn perform_alignment(&mut self, del: f64, ext: f64, matrix: &Array2<f64>, heuristics: Option<Heuristics>)
        -> Result<AlignmentResult<T, Alignment<T>>> {
        let (n, m) = (self.query.len(), self.target.len());
        let want = WANT_MATRICES.load(std::sync::atomic::Ordering::Relaxed);
        let q: Vec<u8> = self.query.iter().map(|r| Into::<usize>::into(*r) as u8).collect();
        let t: Vec<u8> = self.target.iter().map(|r| Into::<usize>::into(*r) as u8).collect();
        let mat = matrix.as_standard_layout().to_owned();          // contiguous rows; row_stride = ncols
        let p = params($sem, del, ext, &mat, heuristics.is_some(),
                       OUT_SCORE | OUT_TRACEBACK | if want { OUT_DIRECTIONS | OUT_H } else { 0 }, Into::<usize>::into(T::blank()) as u8);
        let mut res = AlnPairResult::default();
        let (mut qa, mut ta) = (vec![0u8; n + m + 2], vec![0u8; n + m + 2]);
        let mut dirs = vec![0u8; if want { (n + 1) * (m + 1) } else { 0 }];
        let mut h = if want { Array2::<f64>::zeros((m + 1, n + 1)) } else { Array2::<f64>::zeros((0, 0)) };
        let null = std::ptr::null_mut();
        check(unsafe { aln_align_pair(ctx(), &p, q.as_ptr(), n, t.as_ptr(), m, &mut res, qa.as_mut_ptr(), ta.as_mut_ptr(),
                                      if want { dirs.as_mut_ptr() } else { null }, if want { h.as_mut_ptr() } else { null as *mut f64 }) })?;
        let len = res.aln_len as usize;
        let dec = |v: &[u8]| v[..len].iter().map(|c| T::from(*c as usize)).collect::<Vec<T>>();
        let coords = if $global { ((1, n), (1, m)) }                            // simple/mod.rs:138
            else { ((res.start_x as usize + 1, res.end_x as usize + 1), (res.start_y as usize + 1, res.end_y as usize + 1)) };   // :255-258
        Ok(AlignmentResult { alignment_matrix: h, direction_matrix: directions(dirs, if want { (m + 1, n + 1) } else { (0, 0) }),
            alignment: Alignment { query: dec(&qa), target: dec(&ta), coords, f: res.f },
            matrix: None, phantom: PhantomData })
    }
}}}
hip_aligner!(HipGlobalAligner, CORE_GLOBAL, true);
hip_aligner!(HipLocalAligner, CORE_LOCAL, false);

/// `PWMAligner` (pwm/mod.rs:9-126): `query` are the rows, the columns are the positions of the 4 x W position-weight matrix.
/// The library hands the numbered half of the alignment back as `u32` column numbers (0 = gap, pwm/mod.rs:86-101).
pub struct HipPWMAligner<T: BioData + Into<usize> + Copy + Eq> { pub query: Vec<T> }

impl<T: BioData + Into<usize> + From<usize> + Copy + Eq> AlignerTrait<T, PWMAlignment<T>> for HipPWMAligner<T> {
    fn from_str_seqs(query: &str, _target: &str) -> Result<Self> { Ok(HipPWMAligner { query: T::str_to_vec(query)? }) }     // pwm/mod.rs:14-22
    fn from_seqs(query: &[T], _target: &[T]) -> Result<Self> { Ok(HipPWMAligner { query: Vec::from(query) }) }              // pwm/mod.rs:24-28
    fn perform_alignment(&mut self, del: f64, ext: f64, matrix: &Array2<f64>, heuristics: Option<Heuristics>)
        -> Result<AlignmentResult<T, PWMAlignment<T>>> {
        let (w, m) = (matrix.ncols(), self.query.len());           // dim = (query.len() + 1, W + 1), pwm/mod.rs:46
        let want = WANT_MATRICES.load(std::sync::atomic::Ordering::Relaxed);
        let t: Vec<u8> = self.query.iter().map(|r| Into::<usize>::into(*r) as u8).collect();
        let mat = matrix.as_standard_layout().to_owned();
        let p = params(PWM_LOCAL, del, ext, &mat, heuristics.is_some(),
                       OUT_SCORE | OUT_TRACEBACK | if want { OUT_DIRECTIONS | OUT_H } else { 0 }, Into::<usize>::into(T::blank()) as u8);
        let mut res = AlnPairResult::default();
        let cap = w + m + 2;
        let mut numbered = vec![0u32; cap];                        // q_aln: uint32 column numbers, 4-byte aligned by construction
        let mut qa = vec![0u8; cap];
        let mut dirs = vec![0u8; if want { (w + 1) * (m + 1) } else { 0 }];
        let mut h = if want { Array2::<f64>::zeros((m + 1, w + 1)) } else { Array2::<f64>::zeros((0, 0)) };
        let null = std::ptr::null_mut();
        // (the `query` argument is ignored for ALN_PWM_LOCAL: N = matrix cols)
        check(unsafe { aln_align_pair(ctx(), &p, std::ptr::null(), w, t.as_ptr(), m, &mut res, numbered.as_mut_ptr() as *mut u8, qa.as_mut_ptr(),
                                      if want { dirs.as_mut_ptr() } else { null }, if want { h.as_mut_ptr() } else { null as *mut f64 }) })?;
        let len = res.aln_len as usize;                            // no duplicated seed pair here (pwm/mod.rs:76-103)
        Ok(AlignmentResult { alignment_matrix: h, direction_matrix: directions(dirs, if want { (m + 1, w + 1) } else { (0, 0) }),
            alignment: PWMAlignment {
                numbered: numbered[..len].iter().map(|c| *c as usize).collect(),
                query: qa[..len].iter().map(|c| T::from(*c as usize)).collect(),
                dim: w,                                                                                                // pwm/mod.rs:113
                coords: ((res.start_x as usize + 1, res.end_x as usize + 1), (res.start_y as usize + 1, res.end_y as usize + 1)),   // :114-117
                f: res.f },
            matrix: None, phantom: PhantomData })
    }
}

/// The batch site of the reference, `calculate_p_value` (statistics/mod.rs:255-286): one query against `targets.len()` shuffled
/// targets, only `alignment.f` is kept.  One call instead of ten threads x 500 aligners; `del` / `ins` / `matrix` as there.
pub fn local_scores<T: BioData + Into<usize> + Copy + Eq>(query: &[T], targets: &[Vec<T>], del: f64, ins: f64, matrix: &Array2<f64>) -> Vec<f64> {
    let mut seqs: Vec<u8> = query.iter().map(|r| Into::<usize>::into(*r) as u8).collect();        // offset 0: the query, shared by every pair
    let (mut t_off, mut t_len) = (Vec::with_capacity(targets.len()), Vec::with_capacity(targets.len()));
    for t in targets {
        t_off.push(seqs.len() as u64);
        t_len.push(t.len() as u64);
        seqs.extend(t.iter().map(|r| Into::<usize>::into(*r) as u8));
    }
    let (q_off, q_len) = (vec![0u64; targets.len()], vec![query.len() as u64; targets.len()]);
    let mut results = vec![AlnPairResult::default(); targets.len()];
    let mat = matrix.as_standard_layout().to_owned();
    let p = params(CORE_LOCAL, del, ins, &mat, false, OUT_SCORE, Into::<usize>::into(T::blank()) as u8);
    let st = unsafe { aln_align_batch(ctx(), &p, seqs.as_ptr(), q_off.as_ptr(), q_len.as_ptr(), t_off.as_ptr(), t_len.as_ptr(), targets.len(),
                                      results.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null()) };
    check(st).expect("aligner_hip batch call");
    results.iter().map(|r| { check(r.status).expect("aligner_hip pair"); r.f }).collect()         // what :273-279 kept
}
