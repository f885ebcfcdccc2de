"""Host-side mirror of the reference's operator interface for the accelerated path.

Same names, argument meaning and error behaviour as ``string_grouper/string_grouper.py`` of
Bergvca/string_grouper v0.7.1 (``match_strings`` :130, ``match_most_similar`` :95,
``group_similar_strings`` :70, ``compute_pairwise_similarities`` :55, ``StringGrouperConfig`` :156,
``StringGrouper`` :224), written from scratch around a device engine: ``fit()`` sends the string
columns to the GPU once, builds the TF-IDF matrices there (K1 + K2), multiplies with fused threshold /
top-n (K3 + K4) and only the match list comes back.  Everything after that (symmetrising, match
frames, groups) is small host work on numpy arrays.

Differences from the reference that a caller can observe, all deliberate:
* the vectoriser is fitted once per ``fit()`` (the reference tokenises the master column three times);
* block counts only steer how the GPU work is cut up; results are identical for every ``n_blocks``
  (the reference's own invariant, test_string_grouper.py:191-336);
* ties at the ``max_n_matches`` cut are resolved by (similarity descending, position ascending).
"""
from __future__ import annotations

import functools
import inspect
import logging
import multiprocessing
import re
from typing import List, NamedTuple, Optional, Tuple, Union
from unicodedata import normalize as _nfkd

import numpy as np
import pandas as pd
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components

from . import _hostops
from . import engine as _engine_mod

logger = logging.getLogger("string_grouper_amd")

# ---- defaults (values fixed by the reference's public contract, string_grouper.py:17-37)
DEFAULT_NGRAM_SIZE: int = 3
DEFAULT_TFIDF_MATRIX_DTYPE: type = np.float64
DEFAULT_REGEX: str = r'[,-./]|\s'
DEFAULT_MAX_N_MATCHES: int = 20
DEFAULT_MIN_SIMILARITY: float = 0.8
DEFAULT_N_PROCESSES: int = multiprocessing.cpu_count() - 1
DEFAULT_IGNORE_CASE: bool = True
DEFAULT_DROP_INDEX: bool = False
DEFAULT_REPLACE_NA: bool = False
DEFAULT_INCLUDE_ZEROES: bool = True
GROUP_REP_CENTROID: str = 'centroid'
GROUP_REP_FIRST: str = 'first'
DEFAULT_GROUP_REP: str = GROUP_REP_CENTROID
DEFAULT_FORCE_SYMMETRIES: bool = True
DEFAULT_N_BLOCKS: Optional[Tuple[int, int]] = None
DEFAULT_NORMALIZE_TO_ASCII: bool = True

# ---- output naming (string_grouper.py:39-49)
DEFAULT_COLUMN_NAME: str = 'side'
DEFAULT_ID_NAME: str = 'id'
LEFT_PREFIX: str = 'left_'
RIGHT_PREFIX: str = 'right_'
MOST_SIMILAR_PREFIX: str = 'most_similar_'
DEFAULT_MASTER_NAME: str = 'master'
DEFAULT_MASTER_ID_NAME: str = f'{DEFAULT_MASTER_NAME}_{DEFAULT_ID_NAME}'
GROUP_REP_PREFIX: str = 'group_rep_'


def _pick_rows(series, positions, default_name, drop_index, mirror):
    """``series`` at ``positions`` as the frames hold it -- the values (and, unless dropped, the index of the picked rows as
    ``reset_index`` would name it), index column first unless ``mirror`` -- without pandas' take + reset_index on the common
    shapes.  Shared by get_matches (both sides) and the group representatives."""
    name = series.name if series.name else default_name
    pos = np.asarray(positions)
    idx = series.index
    if drop_index or (idx.nlevels == 1 and name not in ('index', 'level_0') and idx.name not in (name,)):
        # the common shapes, without pandas' take + reset_index (which copy every column twice
        # at millions of rows): gather values -- and the index, as reset_index would name it --
        # with numpy and hand the columns over as they are
        if series.dtype == object:
            # numpy gather of the string pointers; handing pandas an object ndarray avoids the per-element
            # missing-value scan that building a Series from a NumpyExtensionArray costs (0.2 s per side at
            # 2 M rows)
            # (round 6: the gather itself on a few host threads, _hostops.take_objects -- the same objects; a numpy
            #  take raises 2 M reference counts one after the other, 22 ms a side at 663 k names)
            values = pd.Series(_hostops.take_objects(series.to_numpy(), pos), name=name, copy=False, dtype=object)
        else:
            values = pd.Series(series.array.take(pos), name=name, copy=False)  # keeps an extension dtype
        if drop_index:
            return values
        # the index values of the picked rows, as an ndarray where there is one to be had: a Series built from an
        # Index copies it (10 ms per side at 2 M rows), RangeIndex.take materialises the range first (6 ms)
        if type(idx) is pd.RangeIndex:
            picked_index = _hostops.affine_i64(pos, idx.start, idx.step)      # (a new array: start + pos * step)
        elif isinstance(idx.dtype, np.dtype) and idx.dtype.kind in 'iufbO':
            picked_index = idx.to_numpy().take(pos)
        else:
            picked_index = idx.take(pos)              # datetimes, categoricals, extension dtypes: pandas' own way
        index_col = pd.Series(picked_index, name='index' if idx.name is None else idx.name, copy=False)
        return _concat_columns([values, index_col] if mirror else [index_col, values])
    named = series if series.name else series.rename(default_name)
    picked = named.iloc[pos].reset_index(drop=drop_index)
    if mirror and isinstance(picked, pd.DataFrame):
        picked = picked[picked.columns[::-1]]
    return picked


class StringGrouperConfig(NamedTuple):
    """Options of a StringGrouper (field names, order and defaults as string_grouper.py:189-202).

    ngram_size            characters per n-gram
    tfidf_matrix_dtype    np.float32 or np.float64
    regex                 pattern whose matches are deleted from every string before n-gramming
    max_n_matches         matches kept per string of the left-hand series
    min_similarity        matches need a cosine similarity strictly above this
    number_of_processes   accepted for compatibility; the multiply runs on the GPU
    ignore_case, ignore_index, include_zeroes, replace_na, group_rep, force_symmetries,
    n_blocks, normalize_to_ascii   as in the reference documentation
    """
    ngram_size: int = DEFAULT_NGRAM_SIZE
    tfidf_matrix_dtype: int = DEFAULT_TFIDF_MATRIX_DTYPE
    regex: str = DEFAULT_REGEX
    max_n_matches: Optional[int] = DEFAULT_MAX_N_MATCHES
    min_similarity: float = DEFAULT_MIN_SIMILARITY
    number_of_processes: int = DEFAULT_N_PROCESSES
    ignore_case: bool = DEFAULT_IGNORE_CASE
    ignore_index: bool = DEFAULT_DROP_INDEX
    include_zeroes: bool = DEFAULT_INCLUDE_ZEROES
    replace_na: bool = DEFAULT_REPLACE_NA
    group_rep: str = DEFAULT_GROUP_REP
    force_symmetries: bool = DEFAULT_FORCE_SYMMETRIES
    n_blocks: Tuple[int, int] = DEFAULT_N_BLOCKS
    normalize_to_ascii: bool = DEFAULT_NORMALIZE_TO_ASCII


class StringGrouperNotFitException(Exception):
    """A result was requested before ``fit()``."""


def _concat_columns(parts):
    """pd.concat(parts, axis=1) that keeps the columns as they are.  Under copy-on-write concat neither
    copies nor consolidates (consolidation re-copies every object pointer at millions of rows); the parts
    are temporaries nobody else references.  pandas >= 3 behaves like that without the option."""
    try:
        with pd.option_context("mode.copy_on_write", True):
            return pd.concat(parts, axis=1)
    except (KeyError, ValueError):          # the option is gone: copy-on-write is the only mode
        return pd.concat(parts, axis=1)


def validate_is_fit(method):
    @functools.wraps(method)
    def guarded(self, *args, **kwargs):
        if not self.is_build:
            raise StringGrouperNotFitException(
                f'{method.__name__} was called before the "fit" function was called.'
                f' Make sure to run fit the StringGrouper first using StringGrouper.fit()')
        return method(self, *args, **kwargs)
    return guarded


# =================================================================================================
# module-level API (string_grouper.py:55-153)
# =================================================================================================
def compute_pairwise_similarities(string_series_1: pd.Series, string_series_2: pd.Series, **kwargs) -> pd.Series:
    """Row-wise similarity of two equally long series."""
    return StringGrouper(string_series_1, string_series_2, **kwargs).dot()


def group_similar_strings(strings_to_group: pd.Series, string_ids: Optional[pd.Series] = None,
                          **kwargs) -> Union[pd.DataFrame, pd.Series]:
    """For every string the representative of its group of similar strings."""
    grouper = StringGrouper(strings_to_group, master_id=string_ids, **kwargs).fit()
    return grouper.get_groups()


def match_most_similar(master: pd.Series, duplicates: pd.Series, master_id: Optional[pd.Series] = None,
                       duplicates_id: Optional[pd.Series] = None, **kwargs) -> Union[pd.DataFrame, pd.Series]:
    """For every string in ``duplicates`` the most similar string of ``master`` (or itself)."""
    kwargs['max_n_matches'] = 1          # string_grouper.py:120
    grouper = StringGrouper(master, duplicates=duplicates, master_id=master_id, duplicates_id=duplicates_id,
                            **kwargs).fit()
    return grouper.get_groups()


def match_strings(master: pd.Series, duplicates: Optional[pd.Series] = None, master_id: Optional[pd.Series] = None,
                  duplicates_id: Optional[pd.Series] = None, **kwargs) -> pd.DataFrame:
    """All pairs of highly similar strings (self-join when ``duplicates`` is None)."""
    grouper = StringGrouper(master, duplicates=duplicates, master_id=master_id, duplicates_id=duplicates_id,
                            **kwargs).fit()
    return grouper.get_matches()


# =================================================================================================
class StringGrouper(object):
    def __init__(self, master: pd.Series, duplicates: Optional[pd.Series] = None,
                 master_id: Optional[pd.Series] = None, duplicates_id: Optional[pd.Series] = None, **kwargs):
        self.is_build = False
        self._master: pd.Series = pd.Series(dtype=object)
        self._duplicates: Optional[pd.Series] = None
        self._master_id: Optional[pd.Series] = None
        self._duplicates_id: Optional[pd.Series] = None
        self._left_Series = self._right_Series = None
        self._matches_list: pd.DataFrame = pd.DataFrame()
        self._true_max_n_matches: int = 0
        self._max_n_matches: int = 0
        self._vectorizer = None
        self._config = StringGrouperConfig(**kwargs)      # TypeError on an unknown option
        self._n_blocks = self._config.n_blocks
        self._n_blocks_guessed = False
        self._set_data(master, duplicates, master_id, duplicates_id)
        self._set_options(**kwargs)

    # The match list is a plain attribute in the reference (tests and add_match / remove_match assign to
    # it).  Here an assignment also drops the device-resident copy of the previous fit(): the reductions
    # over the list (K7, K8) must never see a list the caller has edited on the host.
    @property
    def _matches_list(self):
        return self.__dict__.get('_matches_list_df')

    @_matches_list.setter
    def _matches_list(self, value):
        self.__dict__['_matches_list_df'] = value
        self._drop_device_matches()

    def _drop_device_matches(self):
        dml = self.__dict__.pop('_device_matches', None)
        if dml is not None:
            dml.free()

    def __getstate__(self):
        """Pickle / deepcopy like the reference's plain-Python object: device handles (the vectoriser's
        vocabulary, the device-resident match list) stay behind; the copy reduces on the host."""
        state = dict(self.__dict__)
        state.pop('_device_matches', None)
        state['_vectorizer'] = None
        return state

    # ------------------------------------------------------------------ data / options
    def _set_data(self, master, duplicates=None, master_id=None, duplicates_id=None):
        self.master = master
        self.duplicates = duplicates
        if not StringGrouper._is_input_data_combination_valid(duplicates, master_id, duplicates_id):
            raise Exception('List of data Series options is invalid')
        StringGrouper._validate_id_data(master, duplicates, master_id, duplicates_id)
        self._master_id = master_id
        self._duplicates_id = duplicates_id
        self._left_Series = self._master
        self._right_Series = self._master if self._duplicates is None else self._duplicates
        self.is_build = False

    def _set_options(self, **kwargs):
        self._config = StringGrouperConfig(**kwargs)
        self._max_n_matches = self._config.max_n_matches
        self._validate_group_rep_specs()
        self._validate_tfidf_matrix_dtype()
        self._validate_replace_na_and_drop()
        StringGrouper._validate_n_blocks(self._config.n_blocks)
        self.is_build = False

    def reset_data(self, master, duplicates=None, master_id=None, duplicates_id=None):
        """Replace the input series, keep the options."""
        self._set_data(master, duplicates, master_id, duplicates_id)

    def clear_data(self):
        self._master = self._duplicates = self._master_id = self._duplicates_id = None
        self._matches_list = None
        self._left_Series = self._right_Series = None
        self.is_build = False

    def update_options(self, **kwargs):
        StringGrouperConfig(**kwargs)                      # validates the names first
        merged = self._config._asdict()
        merged.update(kwargs)
        self._set_options(**merged)

    @property
    def master(self):
        return self._master

    @master.setter
    def master(self, value):
        if not StringGrouper._is_series_of_strings(value):
            raise TypeError('Master input does not consist of pandas.Series containing only Strings')
        self._master = value

    @property
    def duplicates(self):
        return self._duplicates

    @duplicates.setter
    def duplicates(self, value):
        if value is not None and not StringGrouper._is_series_of_strings(value):
            raise TypeError('Duplicates input does not consist of pandas.Series containing only Strings')
        self._duplicates = value

    # ------------------------------------------------------------------ hot path, host view
    def n_grams(self, string: str) -> List[str]:
        """The analyzer of the reference (string_grouper.py:365-378), host version for inspection.
        The device tokeniser (K1) produces the same n-grams; tests compare the two."""
        cfg = self._config
        if cfg.ignore_case and string is not None:
            string = string.lower()
        if cfg.normalize_to_ascii:
            string = _nfkd('NFKD', string).encode('ASCII', 'ignore').decode()
        string = re.sub(cfg.regex, r'', string)
        n = cfg.ngram_size
        return [string[i:i + n] for i in range(len(string) - n + 1)]

    # ---- the reference's vectoriser helpers (string_grouper.py:305-308, :699-707).  The reference fits an sklearn
    #      vectoriser when the instance is built and again inside fit(); here vocabulary, idf and both matrices come out
    #      of ONE pass on the device inside fit() (_tfidf_on_engine), so these two only exist for callers of the private
    #      surface: they run that pass and hand back the fitted vectoriser.
    def _fit_vectorizer(self):
        self._tfidf_on_engine()
        return self._vectorizer

    def _build_corpus(self):
        self._vectorizer = self._fit_vectorizer()
        self.is_build = False

    def _tfidf_on_engine(self):
        cfg = self._config
        eng = _engine_mod.get_engine()
        A, B, vec = eng.tfidf(self._master, self._duplicates, cfg.ngram_size, cfg.regex, cfg.ignore_case,
                              cfg.normalize_to_ascii, cfg.tfidf_matrix_dtype)
        self._vectorizer = vec
        return A, B

    def _get_tf_idf_matrices(self) -> Tuple[sp.csr_matrix, sp.csr_matrix]:
        """(master matrix, duplicate matrix) as scipy CSR; the same object twice for a self-join."""
        A, B = self._tfidf_on_engine()
        a = A.to_scipy()
        return a, (a if B is A else B.to_scipy())

    def _build_matches(self, master_matrix, duplicate_matrix, n_blocks: Optional[Tuple[int, int]]) -> sp.csr_matrix:
        """Thresholded top-n cosine similarities, rows = master strings (string_grouper.py:709-752).
        Accepts matrices on the device or scipy matrices."""
        eng = _engine_mod.get_engine()
        A = eng.wrap(master_matrix)
        B = A if duplicate_matrix is master_matrix else eng.wrap(duplicate_matrix)
        top_n, thr = self._max_n_matches, self._config.min_similarity
        if n_blocks is None:
            return eng.topn_multiply(A, B, top_n, thr)
        if getattr(self, '_n_blocks_guessed', False) or tuple(n_blocks) == (1, 1):
            # the guessed split only exists to keep a CPU accumulator in cache; one device multiply
            # gives the identical result (cast as the reference's vstack(dtype=float64) does)
            return eng.topn_multiply(A, B, top_n, thr).astype(np.float64)
        return eng.topn_multiply_blocked(A, B, tuple(n_blocks), top_n, thr)

    def fit(self):
        """Compute the match list."""
        master_matrix, duplicate_matrix = self._tfidf_on_engine()
        guess = (max(1, round(len(self._left_Series) / 1e6)), max(1, round(len(self._right_Series) / 4e3)))
        self._n_blocks_guessed = False
        if self._n_blocks is None:
            if guess != (1, 1):
                logger.info("n_blocks parameter is not set; reference-equivalent split would be n_blocks = (%d,%d)",
                            guess[0], guess[1])
            self._n_blocks = guess
            self._n_blocks_guessed = True
        if self._can_fuse_on_device():
            # one device pipeline from the TF-IDF matrices to the match list: multiply, diagonal := 1,
            # symmetrise, compaction (K3 + K4 + K6); only (master_side, dupe_side, similarity) comes back
            eng = _engine_mod.get_engine()
            fix = bool(self._config.force_symmetries and self._duplicates is None)
            keep = 'keep_on_device' in inspect.signature(eng.match_list).parameters   # engine doubles may lack it
            try:
                out = eng.match_list(master_matrix, duplicate_matrix, self._max_n_matches, self._config.min_similarity,
                                     fix, **({'keep_on_device': True} if keep else {}))
            except OverflowError:
                # the one exception the reference's fit() handles (string_grouper.py:397-413): fall back to the
                # block-wise multiply with the reference's own guess of the split
                logger.warning("An OverflowError occurred but is being handled: the input is split into "
                               "n_blocks = (%d, %d) and processed block-wise", guess[0], guess[1])
                out = None
            if out is None:
                self._n_blocks_guessed = False
                matches = self._build_matches(master_matrix, duplicate_matrix, guess if guess != (1, 1) else (2, 1))
                self._true_max_n_matches = int(np.diff(matches.indptr).max()) if matches.shape[0] else 0
                if fix:
                    matches = StringGrouper._symmetrize_matrix(StringGrouper._fix_diagonal(matches))
                self._matches_list = self._get_matches_list(matches)
                self.is_build = True
                return self
            rows, cols, sims, self._true_max_n_matches = out[:4]
            self._matches_list = _concat_columns(      # columns kept as they are: no re-copy
                [pd.Series(rows, name='master_side', copy=False), pd.Series(cols, name='dupe_side', copy=False),
                 pd.Series(sims if sims.dtype == np.float64 else _hostops.widen(sims, np.float64), name='similarity', copy=False)])
            # the same list stays in HBM for get_groups(): best master per duplicate (K7) / group
            # representatives (K8) come back as one int32 per string
            if len(out) > 4:
                self.__dict__['_device_matches'] = out[4]
            self.is_build = True
            return self
        if self._n_blocks == (1, 1):
            try:
                matches = self._build_matches(master_matrix, duplicate_matrix, self._n_blocks)
            except OverflowError:
                logger.warning("An OverflowError occurred but is being handled: the input is split into "
                               "n_blocks = (%d, %d) and processed block-wise", guess[0], guess[1])
                self._n_blocks_guessed = False
                matches = self._build_matches(master_matrix, duplicate_matrix, guess)
        else:
            matches = self._build_matches(master_matrix, duplicate_matrix, self._n_blocks)

        self._true_max_n_matches = int(np.diff(matches.indptr).max()) if matches.shape[0] else 0
        if self._config.force_symmetries and self._duplicates is None:
            matches = StringGrouper._fix_diagonal(matches)
            matches = StringGrouper._symmetrize_matrix(matches)
        self._matches_list = self._get_matches_list(matches)
        self.is_build = True
        return self

    def _can_fuse_on_device(self) -> bool:
        """The fused device tail is used unless a caller replaced one of the hooks the reference exposes
        (tests patch ``_build_matches`` / ``_fix_diagonal`` / ``_symmetrize_matrix``), asked for explicit
        blocks, or the engine is a test double."""
        eng = _engine_mod.get_engine()
        return (hasattr(eng, 'match_list')
                and '_build_matches' not in self.__dict__
                and type(self)._build_matches is _ORIGINAL_HOOKS[0]
                and StringGrouper.__dict__['_fix_diagonal'] is _ORIGINAL_HOOKS[1]
                and StringGrouper.__dict__['_symmetrize_matrix'] is _ORIGINAL_HOOKS[2]
                and self._n_blocks is not None
                and (self._n_blocks_guessed or tuple(self._n_blocks) == (1, 1)))

    def dot(self) -> pd.Series:
        """Row-wise similarity between master and duplicates."""
        if len(self._master) != len(self._duplicates):
            raise Exception("To perform this function, both input Series must have the same length.")
        eng = _engine_mod.get_engine()
        if hasattr(eng, 'rowwise_dot'):      # on the device (K9): only the similarities come back
            A, B = self._tfidf_on_engine()
            sims = eng.rowwise_dot(A, B)
        else:
            a, b = self._get_tf_idf_matrices()
            sims = np.asarray(a.multiply(b).sum(axis=1)).squeeze(axis=1)
        return pd.Series(sims, name='similarity', index=self._master.index)

    # ------------------------------------------------------------------ post-processing
    @staticmethod
    def _fix_diagonal(m):
        """Every string matches itself with similarity exactly 1 (string_grouper.py:954-958): every
        diagonal entry is set to 1, also for rows that had none."""
        coo = sp.coo_matrix(m)
        n = coo.shape[0]
        off = coo.row != coo.col
        diag = np.arange(n, dtype=coo.row.dtype)
        rows = np.concatenate([coo.row[off], diag])
        cols = np.concatenate([coo.col[off], diag])
        vals = np.concatenate([coo.data[off], np.ones(n, dtype=coo.data.dtype)])
        return sp.coo_matrix((vals, (rows, cols)), shape=coo.shape)

    @staticmethod
    def _symmetrize_matrix(m_symmetric):
        """If (r, c) is stored so is (c, r) with the same value (string_grouper.py:960-964); rows come
        back sorted by column, as the reference's lil round trip leaves them."""
        coo = sp.coo_matrix(m_symmetric)
        n_cols = np.int64(coo.shape[1])
        r = coo.row.astype(np.int64)
        c = coo.col.astype(np.int64)
        keys = np.concatenate([r * n_cols + c, c * n_cols + r])       # stored entries first: they win
        vals = np.concatenate([coo.data, coo.data])
        uniq, first = np.unique(keys, return_index=True)
        rows = uniq // n_cols
        indptr = np.zeros(coo.shape[0] + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=coo.shape[0]), out=indptr[1:])
        idx_dtype = np.int32 if max(coo.shape[1], len(uniq)) < 2 ** 31 else np.int64
        out = sp.csr_matrix((vals[first], (uniq % n_cols).astype(idx_dtype), indptr.astype(idx_dtype)), shape=coo.shape)
        out.has_sorted_indices = True
        return out

    def _get_matches_list(self, matches) -> pd.DataFrame:
        matches = sp.csr_matrix(matches)
        rows = np.repeat(np.arange(matches.shape[0], dtype=np.int64), np.diff(matches.indptr))
        return pd.DataFrame({'master_side': rows, 'dupe_side': matches.indices.astype(np.int64),
                             'similarity': matches.data})

    def _get_non_matches_list(self) -> pd.DataFrame:
        """All pairs that are not in the match list, with similarity 0 (string_grouper.py:765-781)."""
        n_m = len(self._master)
        n_d = len(self._master if self._duplicates is None else self._duplicates)
        present = np.zeros((n_m, n_d), dtype=bool)
        present[self._matches_list.master_side.to_numpy(), self._matches_list.dupe_side.to_numpy()] = True
        ms, ds = np.nonzero(~present)
        if len(ms) == 0:
            return pd.DataFrame()
        if self._max_n_matches < self._true_max_n_matches:
            raise Exception(f'\nERROR: Cannot return zero-similarity matches since \n'
                            f'\t\t max_n_matches={self._max_n_matches} is too small!\n'
                            f'\t\t Try setting max_n_matches={self._true_max_n_matches} (the \n'
                            f'\t\t true maximum number of matches over all strings in master)\n'
                            f'\t\t or greater or do not set this kwarg at all.')
        return pd.DataFrame({'master_side': ms.astype(np.int64), 'dupe_side': ds.astype(np.int64), 'similarity': 0})

    @validate_is_fit
    def get_matches(self, ignore_index: Optional[bool] = None, include_zeroes: Optional[bool] = None) -> pd.DataFrame:
        """The match list as a frame: left strings (+ids, +index), similarity, right strings."""
        if ignore_index is None:
            ignore_index = self._config.ignore_index
        if include_zeroes is None:
            include_zeroes = self._config.include_zeroes
        pairs = self._matches_list
        if self._config.min_similarity <= 0 and include_zeroes:
            missing = self._get_non_matches_list()
            if not missing.empty:
                pairs = pd.concat([pairs, missing], axis=0, ignore_index=True)

        right_source = self._master if self._duplicates is None else self._duplicates

        def prefixed(obj, prefix):
            if isinstance(obj, pd.DataFrame):
                return obj.rename(columns={c: f"{prefix}{c}" for c in obj.columns}, copy=False)
            return obj.rename(f"{prefix}{obj.name}", copy=False)

        left = _pick_rows(self._master, pairs.master_side, DEFAULT_COLUMN_NAME, ignore_index, False)
        right = _pick_rows(right_source, pairs.dupe_side, DEFAULT_COLUMN_NAME, ignore_index, True)
        similarity = pairs.similarity.reset_index(drop=True)
        if self._master_id is None:
            parts = [prefixed(left, LEFT_PREFIX), similarity, prefixed(right, RIGHT_PREFIX)]
        else:
            right_ids = self._master_id if self._duplicates is None else self._duplicates_id
            left_id = _pick_rows(self._master_id, pairs.master_side, DEFAULT_ID_NAME, True, False)
            right_id = _pick_rows(right_ids, pairs.dupe_side, DEFAULT_ID_NAME, True, True)
            parts = [prefixed(left, LEFT_PREFIX), prefixed(left_id, LEFT_PREFIX), similarity,
                     prefixed(right_id, RIGHT_PREFIX), prefixed(right, RIGHT_PREFIX)]
        return _concat_columns(parts)

    @validate_is_fit
    def get_groups(self, ignore_index: Optional[bool] = None,
                   replace_na: Optional[bool] = None) -> Union[pd.DataFrame, pd.Series]:
        """Self-join: the group representative of every string.  Two series: for every duplicate the
        most similar master string."""
        if ignore_index is None:
            ignore_index = self._config.ignore_index
        if self._duplicates is None:
            return self._deduplicate(ignore_index=ignore_index)
        if replace_na is None:
            replace_na = self._config.replace_na
        return self._get_nearest_matches(ignore_index=ignore_index, replace_na=replace_na)

    # ---- methods that re-run with new data (string_grouper.py:546-644)
    def match_strings(self, master, duplicates=None, master_id=None, duplicates_id=None, **kwargs) -> pd.DataFrame:
        self.reset_data(master, duplicates, master_id, duplicates_id)
        self.update_options(**kwargs)
        return self.fit().get_matches()

    def match_most_similar(self, master, duplicates, master_id=None, duplicates_id=None, **kwargs):
        self.reset_data(master, duplicates, master_id, duplicates_id)
        self.update_options(**kwargs)
        return self.fit().get_groups()

    def group_similar_strings(self, strings_to_group, string_ids=None, **kwargs):
        self.reset_data(strings_to_group, master_id=string_ids)
        self.update_options(**kwargs)
        return self.fit().get_groups()

    def compute_pairwise_similarities(self, string_series_1, string_series_2, **kwargs) -> pd.Series:
        self.reset_data(string_series_1, string_series_2)
        self.update_options(**kwargs)
        return self.dot()

    # ---- manual edits of the match list (string_grouper.py:646-683)
    @validate_is_fit
    def add_match(self, master_side: str, dupe_side: str) -> 'StringGrouper':
        m_idx, d_idx = self._get_indices_of(master_side, dupe_side)
        earlier = self._matches_list.master_side[self._matches_list.dupe_side.isin(d_idx)]
        d_idx = pd.concat([d_idx, earlier]).drop_duplicates()
        new_pairs = StringGrouper._cross_join(d_idx, m_idx, [1])
        if self._duplicates is None:
            new_pairs = StringGrouper._make_symmetric(new_pairs)
        self._matches_list = pd.concat([self._matches_list.drop_duplicates(), new_pairs], ignore_index=True)
        return self

    @validate_is_fit
    def remove_match(self, master_side: str, dupe_side: str) -> 'StringGrouper':
        m_idx, d_idx = self._get_indices_of(master_side, dupe_side)
        if self._duplicates is None:       # symmetric: drop both directions
            m_idx = pd.concat([m_idx, d_idx])
            d_idx = m_idx
        hit = self._matches_list.master_side.isin(m_idx) & self._matches_list.dupe_side.isin(d_idx)
        self._matches_list = self._matches_list[~hit]
        return self

    @staticmethod
    def _make_symmetric(new_matches: pd.DataFrame) -> pd.DataFrame:
        flipped = pd.DataFrame({'master_side': new_matches.dupe_side, 'dupe_side': new_matches.master_side,
                                'similarity': new_matches.similarity})
        return pd.concat([new_matches, flipped])

    @staticmethod
    def _cross_join(dupe_indices, master_indices, similarities) -> pd.DataFrame:
        """Every (master index, duplicate index, similarity) combination as match-list rows (string_grouper.py:974-978)."""
        grid = pd.MultiIndex.from_product([master_indices, dupe_indices, similarities],
                                          names=['master_side', 'dupe_side', 'similarity'])
        return pd.DataFrame(index=grid).reset_index()

    @staticmethod
    def _validate_strings_exist(master_side, dupe_side, master_strings, dupe_strings):
        """ValueError unless both strings occur in their columns (string_grouper.py:981-985)."""
        if not master_strings.isin([master_side]).any():
            raise ValueError(f'{master_side} not found in StringGrouper string series')
        if not dupe_strings.isin([dupe_side]).any():
            raise ValueError(f'{dupe_side} not found in StringGrouper dupe string series')

    def _get_indices_of(self, master_side: str, dupe_side: str) -> Tuple[pd.Series, pd.Series]:
        m_strings = self._master
        d_strings = self._master if self._duplicates is None else self._duplicates
        self._validate_strings_exist(master_side, dupe_side, m_strings, d_strings)
        m_idx = m_strings[m_strings == master_side].index.to_series().reset_index(drop=True)
        d_idx = d_strings[d_strings == dupe_side].index.to_series().reset_index(drop=True)
        return m_idx, d_idx

    # ---- match_most_similar result (string_grouper.py:783-849)
    def _best_master_positions(self) -> np.ndarray:
        """For every duplicate (by position) the position of its best master, -1 when it has no match: the highest
        similarity wins, the lowest master position among equals (string_grouper.py:803-807)."""
        n_dupes = len(self._duplicates)
        dml = self.__dict__.get('_device_matches')
        if dml is not None:     # reduced on the device (K7): one int32 per duplicate crosses PCIe
            return dml.best_master().astype(np.int64)
        # the list was edited on the host (add_match / remove_match) or built there
        ml = self._matches_list
        ms, ds, sim = ml.master_side.to_numpy(), ml.dupe_side.to_numpy(), ml.similarity.to_numpy()
        order = np.lexsort((ms, -sim, ds))             # per duplicate: best similarity first, then lowest master
        ds_sorted = ds[order]
        first = np.ones(len(order), dtype=bool)
        first[1:] = ds_sorted[1:] != ds_sorted[:-1]
        best = np.full(n_dupes, -1, dtype=np.int64)
        best[ds_sorted[first]] = ms[order][first]
        return best

    @staticmethod
    def _fill_from(column: pd.Series, rows: np.ndarray, donor: pd.Series, like_dtype, donor_dtype) -> pd.Series:
        """``column`` with the entries at ``rows`` (a mask) taken from ``donor`` -- a join that finds no partner leaves a
        hole and widens the column (integers become floats, booleans objects); where the donor is of the column's original
        type the original type is put back, as the reference does (string_grouper.py:821-826, :840-843).  No element of a
        type the column cannot hold is ever assigned into it: the two are combined, which widens without complaint."""
        filled = column.where(~rows, donor) if rows.any() else column
        if filled.dtype != like_dtype and donor_dtype == like_dtype:
            filled = filled.astype(like_dtype)
        return filled

    def _get_nearest_matches(self, ignore_index=False, replace_na=False) -> Union[pd.DataFrame, pd.Series]:
        """One row per duplicate, in the duplicates' order and under their index: the best master's string (and id, and
        index levels unless ``ignore_index``), or the duplicate's own where it has no match (its index levels only with
        ``replace_na``).  Assembled by POSITION -- the best master of duplicate d is a row number of the master table --
        instead of the reference's chain of key joins; the frames are the reference's, dtype for dtype
        (tests/test_host_api.py: fuzz against the mounted reference)."""
        prefix = MOST_SIMILAR_PREFIX
        name_col = f'{prefix}{self._master.name if self._master.name else DEFAULT_MASTER_NAME}'
        # both sides as positional tables: the index levels (unless dropped) to the left of the strings.  The MASTER's table is
        # never built in full (round 6: a copy of 663 k strings and their index for the 165 k rows that are picked from it):
        # `m_tbl` is its empty head -- the columns and their dtypes, which is what the rest of this function asks of it -- and
        # the rows of the best masters are gathered directly.
        best = self._best_master_positions()
        lonely = best < 0                                   # duplicates without a match
        m_named = self._master.rename(name_col, copy=False)
        m_tbl = m_named.iloc[:0].reset_index(drop=ignore_index)
        d_tbl = self._duplicates.rename('duplicates', copy=False).reset_index(drop=ignore_index)
        if len(self._master) > 0:
            safe = np.where(lonely, 0, best)
            picked = _pick_rows(m_named, safe, name_col, ignore_index, False)
        else:                                               # (nothing to gather from: every duplicate is lonely)
            safe = None
            picked = m_tbl.reindex(pd.Index(best)).reset_index(drop=True)
        if isinstance(d_tbl, pd.DataFrame):
            m_tbl = m_tbl.rename(columns={c: f'{prefix}{c}' for c in m_tbl.columns if str(c) != name_col})
            if isinstance(picked, pd.DataFrame):
                picked = picked.rename(columns={c: f'{prefix}{c}' for c in picked.columns if str(c) != name_col}, copy=False)
        id_col = None
        if self._master_id is not None:
            id_col = f'{prefix}{self._master_id.name if self._master_id.name else DEFAULT_MASTER_ID_NAME}'
            id_named = self._master_id.rename(id_col, copy=False)
            m_tbl = pd.concat([m_tbl, id_named.iloc[:0].reset_index(drop=True)], axis=1)
            picked_id = _pick_rows(id_named, safe, id_col, True, False) if safe is not None \
                else id_named.iloc[:0].reindex(pd.Index(best)).reset_index(drop=True)
            picked = pd.concat([picked, picked_id], axis=1)
            d_tbl = pd.concat([d_tbl, self._duplicates_id.rename('duplicates_id', copy=False).reset_index(drop=True)], axis=1)
        m_frame = m_tbl.to_frame() if isinstance(m_tbl, pd.Series) else m_tbl
        d_frame = d_tbl.to_frame() if isinstance(d_tbl, pd.Series) else d_tbl
        picked = picked.to_frame() if isinstance(picked, pd.Series) else picked
        # -1 is no row of the master table: those rows come out empty (and widen their columns, as a reindex does)
        if safe is not None and lonely.any():
            picked = pd.concat([picked[c].where(~lonely) for c in picked.columns], axis=1)

        picked[name_col] = self._fill_from(picked[name_col], lonely, d_frame['duplicates'], m_frame[name_col].dtype,
                                           m_frame[name_col].dtype)
        if id_col is not None:
            picked[id_col] = self._fill_from(picked[id_col], lonely, d_frame['duplicates_id'], self._master_id.dtype,
                                             self._duplicates_id.dtype)
        wanted = [name_col] if id_col is None else [id_col, name_col]
        level_cols = [c for c in m_tbl.columns if c not in wanted] if isinstance(m_tbl, pd.DataFrame) else []
        if replace_na:
            donors = [c for c in d_frame.columns if str(c) != 'duplicates']
            if len(donors) != len(level_cols) and (len(level_cols) > 0 or lonely.any()):
                # (the reference assigns the duplicates' columns -- the id column among them -- to the master's index
                #  columns as one block, string_grouper.py:836-838: with ids the widths differ and pandas refuses)
                picked.loc[lonely, level_cols] = d_frame.loc[lonely, donors].values
            for m_c, d_c in zip(level_cols, donors):
                picked[m_c] = self._fill_from(picked[m_c], lonely, d_frame[d_c], m_frame[m_c].dtype, d_frame[d_c].dtype)
        output = picked[level_cols + wanted]
        output.index = self._duplicates.index
        return output.squeeze(axis=1)

    # ---- group_similar_strings result (string_grouper.py:851-904)
    def _deduplicate(self, ignore_index=False) -> Union[pd.DataFrame, pd.Series]:
        n = len(self._master)
        dml = self.__dict__.get('_device_matches')
        if dml is not None:     # connected components + representatives on the device (K8)
            rep = dml.group_reps(self._config.group_rep == GROUP_REP_CENTROID).astype(np.int64)
        else:
            rep = self._group_reps_on_host(n)

        prefix = GROUP_REP_PREFIX
        label = f'{prefix}{self._master.name}' if self._master.name else prefix[:-1]
        # (round 6: the strings of 663 k representatives through the threaded gather of get_matches' sides -- pandas' take of an
        #  object column was 16 of this function's 39 ms; the frame is the one iloc + rename + reset_index builds)
        output = _pick_rows(self._master.rename(label, copy=False), rep, label, ignore_index, False)
        if isinstance(output, pd.DataFrame):
            output.rename(columns={c: f'{prefix}{c}' for c in output.columns if str(c) != label}, inplace=True)
        if self._master_id is not None:
            id_label = f'{prefix}{self._master_id.name if self._master_id.name else DEFAULT_ID_NAME}'
            output_id = self._master_id.iloc[rep].rename(id_label).reset_index(drop=True)
            output = pd.concat([output_id, output], axis=1)
        output.index = self._master.index
        return output

    def _group_reps_on_host(self, n: int) -> np.ndarray:
        """Representative of every string's group from the host copy of the match list (used when the
        list was edited on the host, or built there)."""
        pairs = self._matches_list
        ms, ds = pairs.master_side.to_numpy(), pairs.dupe_side.to_numpy()
        graph = sp.csr_matrix((np.full(len(pairs), 1), (ms, ds)), shape=(n, n))
        _, labels = connected_components(csgraph=graph, directed=True)
        if self._config.group_rep == GROUP_REP_CENTROID:
            graph.data = pairs['similarity'].to_numpy()
            weight = np.asarray(graph.sum(axis=1)).squeeze(axis=1)
            # per group the member with the largest similarity aggregate, first one on ties
            order = np.lexsort((np.arange(n), -weight, labels))
        else:
            order = np.lexsort((np.arange(n), labels))        # first member of each group
        lab_sorted = labels[order]
        head = np.ones(n, dtype=bool)
        head[1:] = lab_sorted[1:] != lab_sorted[:-1]
        rep_of_label = np.empty(labels.max() + 1 if n else 0, dtype=np.int64)
        rep_of_label[lab_sorted[head]] = order[head]
        return rep_of_label[labels]

    # ---- validation (string_grouper.py:916-1010)
    def _validate_group_rep_specs(self):
        allowed = (GROUP_REP_FIRST, GROUP_REP_CENTROID)
        if self._config.group_rep not in allowed:
            raise Exception(f"Invalid option value for group_rep. The only permitted values are\n {allowed}")

    def _validate_tfidf_matrix_dtype(self):
        allowed = (np.float32, np.float64)
        if self._config.tfidf_matrix_dtype not in allowed:
            raise Exception(f"Invalid option value for tfidf_matrix_dtype. The only permitted values are\n {allowed}")

    def _validate_replace_na_and_drop(self):
        if self._config.ignore_index and self._config.replace_na:
            raise Exception("replace_na can only be set to True when ignore_index=False.")
        if self._config.replace_na and self._master.index.nlevels != self._duplicates.index.nlevels:
            raise Exception("replace_na=True: Cannot replace NaN values of index-columns with the values of another "
                            "index if the number of index-levels does not equal the number of index-columns.")

    @staticmethod
    def _validate_n_blocks(n_blocks):
        if n_blocks is None:
            return
        ok = isinstance(n_blocks, tuple) and len(n_blocks) == 2 and \
            all(isinstance(b, int) and b >= 1 for b in n_blocks)
        if not ok:
            raise Exception("Invalid option value for parameter n_blocks: "
                            "n_blocks must be None or a tuple of 2 integers greater than 0.")

    @staticmethod
    def _is_series_of_strings(series_to_test) -> bool:
        if not isinstance(series_to_test, pd.Series):
            return False
        # the reference tests every element with isinstance(x, str) (string_grouper.py:351-362); pandas' C-level type
        # inference answers the same question ("string" only when every element is a str, no missing values)
        if len(series_to_test) == 0:
            return True                       # nothing in it that is not a str (the reference: any() of nothing)
        if series_to_test.dtype == object and _hostops.all_exact_str(series_to_test.to_numpy()):
            return True                       # (round 6: the type test on a few host threads, 7 -> 1 ms at 663 k; anything but str: below)
        kind = pd.api.types.infer_dtype(series_to_test, skipna=False)
        if kind == "string":
            # an extension string dtype may hold pd.NA, which the inference does not report
            return series_to_test.dtype == object or not bool(series_to_test.isna().any())
        if kind == "empty":
            return len(series_to_test) == 0
        if kind in ("mixed", "mixed-integer", "floating", "integer", "boolean", "bytes", "decimal", "complex",
                    "datetime64", "datetime", "date", "timedelta64", "timedelta", "time", "period", "interval"):
            return False                      # some element is not a str (an object column of anything but str included)
        # whatever else the inference calls it ("categorical", "unknown-array", ...): the reference's own element-wise
        # test -- a categorical Series of str passes it
        return all(isinstance(x, str) for x in series_to_test.to_numpy(dtype=object))

    @staticmethod
    def _is_input_data_combination_valid(duplicates, master_id, duplicates_id) -> bool:
        if duplicates is None:
            return duplicates_id is None
        return (master_id is None) == (duplicates_id is None)

    @staticmethod
    def _validate_id_data(master, duplicates, master_id, duplicates_id):
        if master_id is not None and len(master) != len(master_id):
            raise Exception('Both master and master_id must be pandas.Series of the same length.')
        if duplicates is not None and duplicates_id is not None and len(duplicates) != len(duplicates_id):
            raise Exception('Both duplicates and duplicates_id must be pandas.Series of the same length.')


_ORIGINAL_HOOKS = (StringGrouper._build_matches, StringGrouper.__dict__['_fix_diagonal'],
                   StringGrouper.__dict__['_symmetrize_matrix'])
