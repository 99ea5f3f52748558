"""aclgpu -- MI355X-native batched ACL-check engine (host-side Python mirror over the C ABI).

Product code never imports anything under oracle/; there is no CPU evaluation path.
"""
from .engine import (AclError, Engine, ITEM_DTYPE, NO_RELATION, OP_CREATE, OP_DELETE, OP_TOUCH, PERM_CONDITIONAL, PERM_HAS, PERM_NO,
                     PERM_UNSPECIFIED, PRE_MUST_MATCH, PRE_MUST_NOT_MATCH, ERR_ALREADY_EXISTS, ERR_DEPTH, ERR_FAILED_PRECONDITION,
                     ERR_INVALID_ARGUMENT, ERR_OUT_OF_RANGE, ERR_CANCELLED, ERR_DEADLINE_EXCEEDED, ERR_RESOURCE_EXHAUSTED, ERR_UNAVAILABLE, WATCH_FROM_NOW)
from .text import format_relationship, parse_relationship
from . import client, _lib

__all__ = ["Engine", "AclError", "client", "parse_relationship", "format_relationship", "ITEM_DTYPE"]
