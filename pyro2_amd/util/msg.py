"""Coloured terminal messages; same call surface as pyro/util/msg.py:20-52
(fail / warning / success / bold)."""
import sys
import traceback

_CODES = {"fail": "\033[31m", "warning": "\033[33m", "success": "\033[32m",
          "bold": "\033[1m"}
_END = "\033[0m"


def _emit(kind, text):
    print(f"{_CODES[kind]}{text}{_END}")


def fail(string):
    """red message + stack; exits with status 1 unless the interpreter is
    interactive (pyro/util/msg.py:20-31)"""
    _emit("fail", string)
    traceback.print_stack()
    if not hasattr(sys, "ps1"):
        sys.exit(1)


def warning(string):
    _emit("warning", string)


def success(string):
    _emit("success", string)


def bold(string):
    _emit("bold", string)
