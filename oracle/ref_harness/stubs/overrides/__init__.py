"""Tests-only stand-in for the `overrides` package (absent here): the decorator is a no-op marker."""


def overrides(method):
    return method
