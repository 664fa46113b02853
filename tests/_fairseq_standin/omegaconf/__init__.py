"""Test double: only `II` (the interpolation marker) is needed by the plugin."""


def II(key):
    return "${" + key + "}"
