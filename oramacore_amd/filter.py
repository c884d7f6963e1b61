"""Host-side materialiser of the `where` filter: `FilterContext::execute_filter` and `calculate_filter`
(src/collection_manager/sides/read/index/filter.rs:33-392) over the mirror's filter fields, into the bitmap the kernels take.

SURVEY §8(f) rank 1: every filtered search — and every search after an uncommitted delete — hands the scan a
`FilterResult<DocumentId>` (a Bloom-filter-backed predicate tree: Filter | Not | and | or, oramacore_lib).  On the GPU the
predicate is an `AllowBitmap` over DocumentIds; this module evaluates the tree EXACTLY (a set, not a Bloom filter: the
reference's `expected_items.max(100_000)` exists "to avoid hash collisions", filter.rs:347) and produces that bitmap.

Semantics restated from the reference, with the line that pins each:
  * a key the index does not hold as a filter field -> the WHOLE (sub)filter is the empty set for that index (:190-196);
    a filter of the wrong type for the field -> empty set (:55-60, 77-82, 113-118);
  * field entries, `and`, `or` and `not` of one level are AND-ed (:186-283); an empty `or` list is the empty set (:241-245);
    a level with nothing in it is the empty set (:277-281);
  * `not` is the complement (FilterResult::Not): documents of the index that do not satisfy the inner filter;
  * uncommitted deletes: `NOT(deleted)` alone when the filter is empty (:352-364), AND-ed otherwise (:379-390); no filter and
    no deletes -> None (no predicate at all);
  * number filters (types.rs Number / NumberFilter; tests/filter.rs:42-300): eq / gt / gte / lt / lte / between (both ends
    inclusive), integers and floats compare by value (gt 5.0 excludes 5); an array value matches when ANY element does
    (tests/filter.rs:427-495);
  * string filters: equality with one of the document's values; only strings of 1..=25 bytes are indexed as filter values
    (EnumStrategy::StringLength(25), write/index/fields.rs:364-388; tests/filter.rs:913-986);
  * bool filters: equality with one of the document's values;
  * date filters (types.rs DateFilter; tests/filter.rs:576-817): the number operators over timestamps; a string value that parses
    as a date makes its field a DATE filter field instead of a string filter one (write/index/mod.rs:811-818).  The reference
    parses with the `dateparser` crate (many formats); this mirror takes RFC 3339 only — what the reference's tests use.
Across the indexes of a collection a key must be a filter field of AT LEAST ONE index, else the search fails with
FilterFieldNotFound (search.rs:435-449): `check_filter_fields`.
"""
from __future__ import annotations

from .embedding_field import AllowBitmap

STRING_FILTER_MAX_BYTES = 25  # EnumStrategy::StringLength(25), write/index/fields.rs:364


class FilterFieldNotFound(KeyError):
    """ReadError::FilterFieldNotFound (search.rs:447)."""


def _values(v):
    return v if isinstance(v, (list, tuple)) else [v]


def _number_matches(x, flt: dict) -> bool:
    (op, arg), = flt.items()
    if op == "eq":
        return x == arg
    if op == "gt":
        return x > arg
    if op == "gte":
        return x >= arg
    if op == "lt":
        return x < arg
    if op == "lte":
        return x <= arg
    if op == "between":
        return arg[0] <= x <= arg[1]
    raise ValueError(f"unknown number filter {op!r}")


def parse_date(s) -> int | None:
    """OramaDate::try_from (types.rs:2122-2130) for RFC 3339 strings: milliseconds since the epoch, None when `s` is no date."""
    if not isinstance(s, str) or len(s) < 10 or not (s[4:5] == "-" and s[7:8] == "-"):
        return None
    from datetime import datetime, timezone
    try:
        t = datetime.fromisoformat(s[:-1] + "+00:00" if s.endswith(("Z", "z")) else s)
    except ValueError:
        return None
    if t.tzinfo is None:
        t = t.replace(tzinfo=timezone.utc)
    return int(round(t.timestamp() * 1000))


def filter_kind(flt) -> str:
    """Filter::{Date, Number, Bool, String} as serde's untagged enum reads the JSON value (types.rs)."""
    if isinstance(flt, bool):
        return "bool"
    if isinstance(flt, str):
        return "string"
    if isinstance(flt, dict) and len(flt) == 1 and next(iter(flt)) in ("eq", "gt", "gte", "lt", "lte", "between"):
        arg = next(iter(flt.values()))
        if all(isinstance(x, str) for x in _values(arg)):
            return "date"
        return "number"
    raise ValueError(f"unsupported filter value {flt!r}")


def all_keys(where: dict | None) -> list[str]:
    """WhereFilter::get_all_keys: every field key of the tree."""
    if not where:
        return []
    out = []
    for k, v in where.items():
        if k == "and" or k == "or":
            for f in v:
                out.extend(all_keys(f))
        elif k == "not":
            out.extend(all_keys(v))
        else:
            out.append(k)
    return out


class FilterContext:
    """FilterContext (filter.rs:296-392) over the mirror's `Index` (or the tests' HostIndex): bool_fields / number_fields /
    string_filter_fields / date_fields (timestamps) are {field name: {DocumentId: value | [values]}}, `document_ids` the live documents,
    `uncommitted_deleted_documents` the deletes since the last commit."""

    def __init__(self, index):
        self.index = index
        self.universe = set(index.document_ids)

    def field_type(self, key: str) -> str | None:
        """path_to_index_id_map.get_filter_field: the type the index holds `key` as, None when it is no filter field."""
        for kind, store in (("bool", self.index.bool_fields), ("number", self.index.number_fields),
                            ("string", self.index.string_filter_fields), ("date", getattr(self.index, "date_fields", {}))):
            if key in store:
                return kind
        return None

    def has_filter_field(self, key: str) -> bool:
        return self.field_type(key) is not None

    # calculate_filter_for_fields, filter.rs:33-151
    def _on_field(self, key: str, flt) -> set:
        kind = self.field_type(key)
        if kind != filter_kind(flt):
            return set()  # "Wrong filter type for ... field - return empty set"
        store = getattr(self.index, {"bool": "bool_fields", "number": "number_fields", "string": "string_filter_fields",
                                     "date": "date_fields"}[kind])[key]
        if kind == "date":  # (values are stored as timestamps; a bound that is no date matches nothing)
            (op, arg), = flt.items()
            bounds = [parse_date(x) for x in _values(arg)]
            if any(b is None for b in bounds):
                return set()
            ts = {op: bounds if op == "between" else bounds[0]}
            return {d for d, v in store.items() if any(_number_matches(x, ts) for x in _values(v)) and d in self.universe}
        if kind == "bool":
            return {d for d, v in store.items() if any(x is flt for x in _values(v)) and d in self.universe}
        if kind == "number":
            return {d for d, v in store.items() if any(_number_matches(x, flt) for x in _values(v)) and d in self.universe}
        return {d for d, v in store.items()
                if any(x == flt and 1 <= len(x.encode("utf-8")) <= STRING_FILTER_MAX_BYTES for x in _values(v)) and d in self.universe}

    # calculate_filter, filter.rs:176-291
    def _calculate(self, where: dict) -> set:
        results = []
        for k, flt in where.items():
            if k in ("and", "or", "not"):
                continue
            if not self.has_filter_field(k):
                return set()  # :190-196
            results.append(self._on_field(k, flt))
        for f in where.get("and") or []:
            results.append(self._calculate(f))
        if where.get("or") is not None:
            parts = [self._calculate(f) for f in where["or"]]
            if not parts:
                return set()
            results.append(set().union(*parts))
        if where.get("not") is not None:
            results.append(self.universe - self._calculate(where["not"]))
        if not results:
            return set()
        out = results.pop()
        for r in results:
            out &= r
        return out

    def execute_filter(self, where: dict | None, n_bits: int | None = None) -> AllowBitmap | None:
        """filter.rs:344-392: None = no predicate; else the bitmap of the documents that pass (deleted ones never do)."""
        deleted = set(getattr(self.index, "uncommitted_deleted_documents", ()) or ())
        if not where:
            if not deleted:
                return None
            allowed = self.universe - deleted
        else:
            allowed = self._calculate(where) - deleted
        if n_bits is None:
            n_bits = (max(self.universe | deleted) + 1) if (self.universe | deleted) else 1
        return AllowBitmap(n_bits, sorted(allowed))

    def allowed_set(self, where: dict | None) -> set | None:
        """The same as a set (the CPU checker's form); None = no predicate."""
        deleted = set(getattr(self.index, "uncommitted_deleted_documents", ()) or ())
        if not where:
            return None if not deleted else self.universe - deleted
        return self._calculate(where) - deleted


def check_filter_fields(indexes, where: dict | None) -> None:
    """search.rs:435-449: every key of the filter must be a filter field of at least one of the searched indexes."""
    ctxs = [FilterContext(i) for i in indexes]
    for k in all_keys(where):
        if not any(c.has_filter_field(k) for c in ctxs):
            raise FilterFieldNotFound(k)
