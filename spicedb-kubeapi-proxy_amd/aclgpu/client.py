"""Host-side mirror of the seam the reference talks to: `v1.PermissionsServiceClient`
(authzed-go v1.10.0; held in proxy.Options.PermissionsClient, reference
pkg/proxy/options.go:82).  Method names, argument meaning and error behaviour follow the
Go interface so the parity tests read like the reference's own tests; the Go shim in
INTEGRATION.md is the same mapping in cgo.

Only the methods the reference invokes are implemented: CheckPermission
(pkg/authz/watch.go:50), CheckBulkPermissions (check.go:48, postfilter.go:134),
LookupResources (lookups.go:65), WriteRelationships (distributedtx/activity.go:60),
ReadRelationships (activity.go:107), DeleteRelationships (e2e/util_test.go:66).  The other
four raise UNIMPLEMENTED, as the shim does.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterator, List, Optional

from .engine import (AclError, Engine, OP_CREATE, OP_DELETE, OP_TOUCH, PERM_HAS, PRE_MUST_MATCH, PRE_MUST_NOT_MATCH)

# authzed.api.v1 enums
PERMISSIONSHIP_UNSPECIFIED, PERMISSIONSHIP_NO_PERMISSION, PERMISSIONSHIP_HAS_PERMISSION, PERMISSIONSHIP_CONDITIONAL_PERMISSION = 0, 1, 2, 3
LOOKUP_PERMISSIONSHIP_HAS_PERMISSION = 1
OPERATION_CREATE, OPERATION_TOUCH, OPERATION_DELETE = OP_CREATE, OP_TOUCH, OP_DELETE
PRECONDITION_MUST_NOT_MATCH, PRECONDITION_MUST_MATCH = PRE_MUST_NOT_MATCH, PRE_MUST_MATCH
CODE_UNIMPLEMENTED = 12


@dataclass
class ObjectReference:
    object_type: str = ""
    object_id: str = ""


@dataclass
class SubjectReference:
    object: ObjectReference = field(default_factory=ObjectReference)
    optional_relation: str = ""


@dataclass
class Relationship:
    resource: ObjectReference = field(default_factory=ObjectReference)
    relation: str = ""
    subject: SubjectReference = field(default_factory=SubjectReference)
    optional_expires_at: int = 0  # unix seconds


@dataclass
class RelationshipUpdate:
    operation: int = OPERATION_TOUCH
    relationship: Relationship = field(default_factory=Relationship)


@dataclass
class SubjectFilter:
    subject_type: str = ""
    optional_subject_id: str = ""
    optional_relation: Optional[str] = None  # None = any, "" = only without relation


@dataclass
class RelationshipFilter:
    resource_type: str = ""
    optional_resource_id: str = ""
    optional_relation: str = ""
    optional_subject_filter: Optional[SubjectFilter] = None


@dataclass
class Precondition:
    operation: int = PRECONDITION_MUST_MATCH
    filter: RelationshipFilter = field(default_factory=RelationshipFilter)


@dataclass
class CheckPermissionRequest:
    resource: ObjectReference = field(default_factory=ObjectReference)
    permission: str = ""
    subject: SubjectReference = field(default_factory=SubjectReference)


@dataclass
class CheckPermissionResponse:
    permissionship: int = PERMISSIONSHIP_UNSPECIFIED
    checked_at: int = 0


CheckBulkPermissionsRequestItem = CheckPermissionRequest


@dataclass
class CheckBulkPermissionsPair:
    request: CheckPermissionRequest = None
    item: Optional[CheckPermissionResponse] = None  # exactly one of item / error is set (check.go:55-62)
    error: Optional[AclError] = None


@dataclass
class CheckBulkPermissionsResponse:
    pairs: List[CheckBulkPermissionsPair] = field(default_factory=list)
    checked_at: int = 0


@dataclass
class LookupResourcesRequest:
    resource_object_type: str = ""
    permission: str = ""
    subject: SubjectReference = field(default_factory=SubjectReference)


@dataclass
class LookupResourcesResponse:
    resource_object_id: str = ""
    permissionship: int = LOOKUP_PERMISSIONSHIP_HAS_PERMISSION
    looked_up_at: int = 0


@dataclass
class WriteRelationshipsResponse:
    written_at: int = 0  # ZedToken: monotonically increasing store revision


@dataclass
class ReadRelationshipsResponse:
    relationship: Relationship = None
    read_at: int = 0


@dataclass
class DeleteRelationshipsResponse:
    deleted_at: int = 0
    relationships_deleted_count: int = 0


def _filter_kwargs(f: RelationshipFilter) -> dict:
    kw = {"rtype": f.resource_type}
    if f.optional_resource_id:
        kw["rid"] = f.optional_resource_id
    if f.optional_relation:
        kw["rel"] = f.optional_relation
    sf = f.optional_subject_filter
    if sf is not None:
        kw["stype"] = sf.subject_type
        if sf.optional_subject_id:
            kw["sid"] = sf.optional_subject_id
        if sf.optional_relation is not None:
            kw["srel"] = sf.optional_relation
    return kw


def _item_tuple(r: CheckPermissionRequest):
    return (r.resource.object_type, r.resource.object_id, r.permission, r.subject.object.object_type, r.subject.object.object_id,
            r.subject.optional_relation)


class PermissionsServiceClient:
    """Drop-in for the embedded-SpiceDB client on the Check/Filter path; all reads are fully consistent."""

    def __init__(self, engine: Engine):
        self.engine = engine

    # --- reads -------------------------------------------------------------
    def CheckPermission(self, req: CheckPermissionRequest) -> CheckPermissionResponse:
        perms, errs = self.engine.check_bulk([_item_tuple(req)])
        if errs[0]:
            raise AclError(errs[0], "check failed")  # the RPC fails: pkg/proxy/options_test.go:101-102
        return CheckPermissionResponse(perms[0], self.engine.revision)

    def CheckBulkPermissions(self, items: List[CheckPermissionRequest]) -> CheckBulkPermissionsResponse:
        perms, errs = self.engine.check_bulk([_item_tuple(i) for i in items])
        rev = self.engine.revision
        pairs = []
        for it, p, e in zip(items, perms, errs):
            if e:
                pairs.append(CheckBulkPermissionsPair(it, None, AclError(e, "check failed")))
            else:
                pairs.append(CheckBulkPermissionsPair(it, CheckPermissionResponse(p, rev), None))
        return CheckBulkPermissionsResponse(pairs, rev)

    def LookupResources(self, req: LookupResourcesRequest) -> Iterator[LookupResourcesResponse]:
        """Stream until exhausted (the Go side reads until io.EOF, lookups.go:75-83)."""
        ids = self.engine.lookup(req.resource_object_type, req.permission, req.subject.object.object_type, req.subject.object.object_id,
                                 req.subject.optional_relation)
        rev = self.engine.revision
        for i in sorted(ids):
            yield LookupResourcesResponse(i, LOOKUP_PERMISSIONSHIP_HAS_PERMISSION, rev)

    def ReadRelationships(self, flt: RelationshipFilter) -> Iterator[ReadRelationshipsResponse]:
        rev = self.engine.revision
        for rt, rid, rel, st, sid, srel, exp in self.engine.read(**_filter_kwargs(flt)):
            yield ReadRelationshipsResponse(Relationship(ObjectReference(rt, rid), rel, SubjectReference(ObjectReference(st, sid), srel), exp), rev)

    # --- writes ------------------------------------------------------------
    def WriteRelationships(self, updates: List[RelationshipUpdate], optional_preconditions: List[Precondition] = ()) -> WriteRelationshipsResponse:
        ups = []
        for u in updates:
            r = u.relationship
            ups.append((u.operation, (r.resource.object_type, r.resource.object_id, r.relation, r.subject.object.object_type, r.subject.object.object_id,
                                      r.subject.optional_relation), r.optional_expires_at))
        pre = [(p.operation, _filter_kwargs(p.filter)) for p in optional_preconditions]
        return WriteRelationshipsResponse(self.engine.write(ups, pre))

    def DeleteRelationships(self, flt: RelationshipFilter) -> DeleteRelationshipsResponse:
        n = self.engine.delete_by_filter(**_filter_kwargs(flt))
        return DeleteRelationshipsResponse(self.engine.revision, n)

    # --- not on the reference's path ----------------------------------------
    def _unimplemented(self, *_a, **_k):
        raise AclError(CODE_UNIMPLEMENTED, "not implemented by the GPU ACL engine (never called by spicedb-kubeapi-proxy)")

    LookupSubjects = ExpandPermissionTree = ExportBulkRelationships = ImportBulkRelationships = _unimplemented


@dataclass
class WatchUpdate:
    operation: int = OPERATION_TOUCH
    relationship: Relationship = field(default_factory=Relationship)


@dataclass
class WatchResponse:
    updates: List[WatchUpdate] = field(default_factory=list)
    changes_through: int = 0


class WatchServiceClient:
    """`v1.WatchServiceClient` mirror (proxy.Options.WatchClient, reference pkg/proxy/options.go:81; consumer
    pkg/authz/watch.go:27-38).  Watch() starts at the head revision, like a Watch without a start cursor; every
    call of the returned poller yields the WatchResponses committed since the previous one (one per revision)."""

    def __init__(self, engine: Engine):
        self.engine = engine

    def Watch(self, optional_object_types=()):
        from .engine import WATCH_FROM_NOW
        types = list(optional_object_types)
        _, cursor = self.engine.watch_poll(WATCH_FROM_NOW, types)
        state = {"cursor": cursor}

        def recv() -> List[WatchResponse]:
            ups, nxt = self.engine.watch_poll(state["cursor"], types)
            state["cursor"] = nxt
            out: List[WatchResponse] = []
            for rev, op, (rt, rid, rel, st, sid, srel) in ups:
                if not out or out[-1].changes_through != rev:
                    out.append(WatchResponse([], rev))
                out[-1].updates.append(WatchUpdate(op, Relationship(ObjectReference(rt, rid), rel, SubjectReference(ObjectReference(st, sid), srel))))
            return out

        return recv


def filter_items_with_bulk_permissions(client: PermissionsServiceClient, resolved: List[Optional[List[CheckPermissionRequest]]]) -> List[bool]:
    """pkg/authz/postfilter.go:58-182 on the engine: `resolved[i]` holds list item i's resolved PostFilter checks
    (None / [] when no template resolved -> the item is kept, postfilter.go:145-150).  One bulk check, keep mask back."""
    items, off = [], [0]
    for checks in resolved:
        for c in checks or []:
            items.append(_item_tuple(c))
        off.append(len(items))
    if not items:
        return [True] * len(resolved)
    return [bool(k) for k in client.engine.check_bulk_keep(items, off)]


class PrefilterResult:
    """prefilterResult of pkg/authz/lookups.go:25-36, backed by the LookupResources bitmap instead of a set of
    NamespacedNames: IsAllowed(object id text) is one hash lookup + one bit test; filter() answers a whole kube list."""

    def __init__(self, engine: Engine, resource_type: str, bitmap):
        self.engine, self.resource_type, self.bitmap = engine, resource_type, bitmap

    @classmethod
    def run_lookup_resources(cls, engine: Engine, req: LookupResourcesRequest) -> "PrefilterResult":
        bm, _ = engine.lookup_bitmap(req.resource_object_type, req.permission, req.subject.object.object_type, req.subject.object.object_id,
                                     req.subject.optional_relation)
        return cls(engine, req.resource_object_type, bm)

    def filter(self, object_ids: List[str]) -> List[bool]:
        return self.engine.bitmap_test_names(self.resource_type, self.bitmap, object_ids).tolist()

    def is_allowed(self, object_id: str) -> bool:
        return self.filter([object_id])[0]

    def filter_response(self, body: bytes, kind: str = "list", id_template: str = "{{namespacedName}}") -> bytes:
        """filterList / filterTable / filterObject (pkg/authz/responsefilterer.go:349-416) on the kube response's bytes: kind = "list" | "table" |
        "object"; a single object outside the allowed set raises AclError("unauthorized", code 7)."""
        k = {"list": Engine.BODY_LIST, "table": Engine.BODY_TABLE, "object": Engine.BODY_OBJECT}[kind]
        return self.engine.prefilter_response(self.resource_type, self.bitmap, id_template, k, body)[0]


def is_allowed(pair: CheckBulkPermissionsPair) -> bool:
    """The reference's allow rule: no error and HAS_PERMISSION (pkg/authz/check.go:55-69)."""
    return pair.error is None and pair.item is not None and pair.item.permissionship == PERM_HAS
