"""The trainer's S3 hooks.  ``setup_s3_args`` is argument plumbing (upstream ``fourm/utils/s3_utils.py:24-27``); the transfer
functions belong to the storage layer, which stays upstream's: they fall through to an upstream checkout when one is configured
(``fourm._upstream``)."""
from .. import _upstream


def setup_s3_args(args):
    """``s3_data_endpoint`` defaults to ``s3_endpoint``."""
    if not getattr(args, "s3_data_endpoint", None):
        args.s3_data_endpoint = getattr(args, "s3_endpoint", "")


_upstream.merge(__name__, globals())
