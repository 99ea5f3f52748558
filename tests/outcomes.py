"""What a call that may fail came to: ("ok", result) or ("err", code) -- so that tests compare the engine and the oracles on FAILURES too.
LookupResources over a permission that holds `&` / `-` fails as a whole when a candidate's forward Check errs (reference pkg/authz/lookups.go:75-83:
the stream ends at the first Recv error); both oracles and the engine must agree on which lookups those are, not only on the id sets."""
ERR_DEPTH = 100


def outcome(fn, *a, **kw):
    try:
        return ("ok", fn(*a, **kw))
    except Exception as ex:  # noqa: BLE001 -- aclgpu.AclError / oracle.orc.OracleError / oracle.pyoracle.LookupFailed: all carry .code
        code = getattr(ex, "code", None)
        if code is None:
            raise
        return ("err", code)
