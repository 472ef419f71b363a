"""Data formats either side of the scoring path, vectorised.

  ratings.dat        user::item::rating::timestamp        src/api/core/data_loader.py:42-110
  *.train.rating     user \\t item [\\t ...]                 src/api/core/data_loader.py:113-146, src/ncf/utils/load_all_data.py:8-32
  *.test.negative    (user, item) \\t neg \\t neg ...        src/ncf/utils/load_all_data.py:36-56 (written by ml_to_ncf.py:98-110)

The reference builds its dense matrices with DataFrame.iterrows() (about 1 M Python iterations at start-up) and parses the
positive pair with eval(); here the files are read in one pass with numpy and nothing from a file is ever evaluated.
"""
import re

import numpy as np

ENCODINGS = ("latin-1", "iso-8859-1", "cp1252", "utf-8")       # src/api/config.py:48


def _read_text(path):
    raw = open(path, "rb").read()
    for enc in ENCODINGS:
        try:
            return raw.decode(enc)
        except UnicodeDecodeError:
            continue
    raise UnicodeDecodeError("data_io", raw, 0, 1, "none of %s decodes %s" % (ENCODINGS, path))


def read_ratings_dat(path):
    """(user int32[n], item int32[n], rating float32[n]) in file order."""
    txt = _read_text(path).replace("::", " ")
    flat = np.array(txt.split(), dtype=np.float64)
    if flat.size % 4:
        raise ValueError("%s: expected 4 fields per line" % path)
    t = flat.reshape(-1, 4)
    return t[:, 0].astype(np.int32), t[:, 1].astype(np.int32), t[:, 2].astype(np.float32)


def rating_triplets(users, items, ratings, user_num, item_num):
    """(user, item, rating) of the entries _create_matrix_from_ratings (data_loader.py:72-110) writes into its matrix, in file
    order: each id space is re-indexed to 0..n-1 in sorted order unless it already is exactly that; entries whose new ids fall
    outside the matrix are dropped.  (A repeated (user, item) is resolved by whoever consumes the triplets: last one wins.)"""
    users = np.asarray(users); items = np.asarray(items); ratings = np.asarray(ratings, np.float32)

    def reindex(ids):
        uniq, inv = np.unique(ids, return_inverse=True)
        return ids.astype(np.int64) if np.array_equal(uniq, np.arange(len(uniq))) else inv.astype(np.int64)

    u = reindex(users); i = reindex(items)
    keep = (u >= 0) & (u < user_num) & (i >= 0) & (i < item_num)
    return u[keep], i[keep], ratings[keep]


def build_rating_matrix(users, items, ratings, user_num, item_num):
    """Dense [user_num, item_num] float32 matrix exactly as _create_matrix_from_ratings (data_loader.py:72-110); for a
    repeated (user, item) the LAST occurrence wins."""
    u, i, r = rating_triplets(users, items, ratings, user_num, item_num)
    mat = np.zeros((user_num, item_num), np.float32)
    mat[u, i] = r                                            # repeated indices: numpy assigns in order, the last one stays
    return mat


def csr_arrays(rows, cols, vals, n_rows, n_cols):
    """(row_ptr int64 [n_rows + 1], col int32 [nnz], val float32 [nnz]) of the matrix the triplets describe: sorted by
    (row, col), the LAST occurrence of a repeated (row, col) kept, zeros dropped.  Host-side half of the sparse device layout
    (SURVEY N2); ops.CsrMatrix.from_triplets uploads it and adds the kernel's segment table."""
    rows = np.asarray(rows, np.int64); cols = np.asarray(cols, np.int64); vals = np.asarray(vals, np.float32)
    key = rows * int(n_cols) + cols
    order = np.argsort(key, kind="stable")
    key = key[order]
    last = np.ones(len(key), bool)
    last[:-1] = key[1:] != key[:-1]
    keep = last & (vals[order] != 0)
    r = rows[order][keep]
    row_ptr = np.zeros(int(n_rows) + 1, np.int64)
    np.cumsum(np.bincount(r, minlength=int(n_rows)), out=row_ptr[1:])
    return row_ptr, cols[order][keep].astype(np.int32), vals[order][keep]


def load_training_matrix(ratings_path, user_num=6040, item_num=3706):
    """data_loader.load_training_matrix for the ratings.dat branch (:29-32)."""
    u, i, r = read_ratings_dat(ratings_path)
    return build_rating_matrix(u, i, r, user_num, item_num)


def load_training_csr(ratings_path, user_num=6040, item_num=3706, item_major=False):
    """The same matrix as load_training_matrix, never densified: CSR arrays of the [users, items] matrix, or with
    item_major=True of its transpose (row i = item i's ratings from all users: the item-based AutoRec's batch,
    inference_engine.py:230)."""
    u, i, r = rating_triplets(*read_ratings_dat(ratings_path), user_num, item_num)
    if item_major:
        return csr_arrays(i, u, r, item_num, user_num)
    return csr_arrays(u, i, r, user_num, item_num)


def read_pairs_tsv(path):
    """int32 [n, 2] (user, item) from a tab-separated file; extra columns are ignored (usecols=[0,1], data_loader.py:117-124)."""
    rows = [ln.split("\t")[:2] for ln in _read_text(path).splitlines() if ln.strip()]
    return np.array(rows, dtype=np.int32).reshape(-1, 2)


def binary_matrix_from_pairs(pairs, user_num=6040, item_num=3706):
    """_load_training_file (data_loader.py:113-146): 1.0 at every in-range (user, item)."""
    pairs = np.asarray(pairs)
    keep = (pairs[:, 0] >= 0) & (pairs[:, 0] < user_num) & (pairs[:, 1] >= 0) & (pairs[:, 1] < item_num)
    mat = np.zeros((user_num, item_num), np.float32)
    mat[pairs[keep, 0], pairs[keep, 1]] = 1.0
    return mat


_POS = re.compile(r"^\(\s*(-?\d+)\s*,\s*(-?\d+)\s*\)$")


_BLANK_LINE = re.compile(r"(?m)^[ \t\r]*$")
_TO_SPACES = {ord("("): " ", ord(")"): " ", ord(","): " ", ord("\t"): " ", ord("\r"): " "}


def _read_test_negative_fast(txt):
    """The well-formed file in one C-speed pass (115,056 cases x 100 ids: 5.3 s line by line, 1.8 s here; 6,040 cases 0.23 -> 0.08 s); None whenever anything
    is unusual -- an empty field, a line that is not '(u, i)<tab>ids', ragged lines, a token that is not an integer -- so
    that the line-by-line reader below decides and words the error.  Same result as that reader on every file it accepts."""
    m = _BLANK_LINE.search(txt)                              # the reference stops at the first empty line
    body = txt[:m.start()] if m else txt
    lines = body.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    n = len(lines)
    if n == 0 or "\t\t" in body or "\t\n" in body or "\t\r" in body or body.endswith("\t"):
        return None
    if body.count("(") != n or body.count(")") != n or body.count(",") != n or "-" in body or "+" in body or "." in body:
        return None
    tabs = lines[0].count("\t")
    if any(not ln.startswith("(") or ln.count("\t") != tabs for ln in lines):
        return None
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # a token that does not parse must not pass as a short read
        try:
            flat = np.fromstring(body.translate(_TO_SPACES), dtype=np.int64, sep=" ")
        except (DeprecationWarning, ValueError):
            return None
    if flat.size != n * (tabs + 2):
        return None
    t = flat.reshape(n, tabs + 2)
    return t[:, 0].astype(np.int32), np.ascontiguousarray(t[:, 1:]).astype(np.int32)


def read_test_negative(path):
    """Cases of the 99-negative protocol: (users int32[c], items int32[c, 1 + n_neg]) with the positive in column 0.

    Same pairs, in the same order, as the `test_data` list of load_all_data (:36-56) -- but the "(u, i)" field is matched
    with a regular expression instead of eval().  Empty negative fields are skipped like the reference does; all lines
    must then have the same number of negatives (the writer always emits 99).
    """
    txt = _read_text(path)
    fast = _read_test_negative_fast(txt)
    if fast is not None:
        return fast
    users, rows = [], []
    for n, line in enumerate(txt.splitlines()):
        if not line.strip():
            break                                              # the reference stops at the first empty line
        arr = line.strip().split("\t")
        m = _POS.match(arr[0].strip())
        if not m:
            raise ValueError("%s:%d: expected '(user, item)', got %r" % (path, n + 1, arr[0]))
        users.append(int(m.group(1)))
        rows.append([int(m.group(2))] + [int(x) for x in arr[1:] if x])
    width = {len(r) for r in rows}
    if len(width) > 1:
        raise ValueError("%s: lines carry different numbers of negatives %s" % (path, sorted(width)))
    if not rows:
        return np.zeros(0, np.int32), np.zeros((0, 0), np.int32)
    return np.array(users, np.int32), np.array(rows, np.int32).reshape(len(rows), -1)


def test_cases_as_pairs(users, items):
    """Flatten cases to the [cases * (1 + n_neg), 2] pair list that load_all_data returns as test_data."""
    users = np.asarray(users); items = np.asarray(items)
    return np.stack([np.repeat(users, items.shape[1]), items.reshape(-1)], axis=1).astype(np.int32)


def write_test_negative(path, users, items):
    """ml_to_ncf._save_test_negative_samples line format (:98-110)."""
    with open(path, "w") as f:
        for u, row in zip(users, items):
            f.write("(%d, %d)\t%s\n" % (int(u), int(row[0]), "\t".join(str(int(x)) for x in row[1:])))
