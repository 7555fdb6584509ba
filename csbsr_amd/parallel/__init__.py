from .reducer import GradBucketReducer  # noqa: F401
