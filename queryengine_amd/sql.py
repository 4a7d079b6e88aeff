"""A small recursive-descent front end for the reference's SQL subset.

The reference parses with ANTLR4 (grammar ``src/main/antlr4/.../Query.g4:1-115``,
builder ``parser/ExpressionAstBuilder.kt:8-132``); antlr4-runtime is not
available here, and the parser is host-side planning outside the hot path
(SURVEY.md section 2 row 21).  This restatement exists to drive the hot path from
SQL text in tests and in bench.py.  It keeps the grammar's precedence, which
follows the order of the alternatives of ``expression`` (Query.g4:27-40):

    unary (- + NOT)  >  * / %  >  + -  >  comparison  >  AND  >  OR

(note that NOT binds tighter than comparison, as in the reference), all binary
operators left-associative, case-insensitive keywords, numeric literals always
DOUBLE, unary minus/plus folded into numeric literals
(ExpressionAstBuilder.kt:104-110).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import List, Optional

from .ast import (AggregationFunction, AggregationFunctionExpression, BooleanLiteralExpression, Expression, Function,
                  FunctionExpression, IdentifierExpression, NumericLiteralExpression, StringLiteralExpression)


class SyntaxException(RuntimeError):
    """parser/ParserHelper.kt:9"""


@dataclass(frozen=True)
class Query:
    """ast/Query.kt:3"""
    select: tuple
    from_: str
    filter: Optional[Expression]
    orderByColumn: Optional[int]


_TOKEN = re.compile(r"""
    (?P<ws>[ \t\r\n]+)
  | (?P<decimal>[0-9]+(?:\.[0-9]+|[eE][-+]?[0-9]+))
  | (?P<integer>[0-9]+)
  | (?P<ident>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<quoted>"(?:[^"]|"")*")
  | (?P<string>'(?:[^']|'')*')
  | (?P<op>==|!=|<>|<=|>=|[-+*/%=<>(),])
""", re.X)

_KEYWORDS = {"SELECT", "FROM", "WHERE", "ORDER", "BY", "NOT", "AND", "OR", "IF", "THEN", "ELSE", "END", "TRUE", "FALSE"}
_COMPARISONS = {"=": Function.CMP_EQ, "==": Function.CMP_EQ, "!=": Function.CMP_NE, "<>": Function.CMP_NE,
                "<": Function.CMP_LT, "<=": Function.CMP_LE, ">=": Function.CMP_GE, ">": Function.CMP_GT}


def _tokenize(text: str):
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise SyntaxException(f"token recognition error at: '{text[pos]}' (position {pos})")
        pos = m.end()
        kind = m.lastgroup
        if kind == "ws":
            continue
        val = m.group(kind)
        if kind == "ident" and val.upper() in _KEYWORDS:
            out.append(("kw", val.upper()))
        else:
            out.append((kind, val))
    out.append(("eof", ""))
    return out


class _Parser:
    def __init__(self, text: str):
        self.toks = _tokenize(text)
        self.i = 0

    def peek(self):
        return self.toks[self.i]

    def take(self):
        t = self.toks[self.i]
        self.i += 1
        return t

    def accept(self, kind, val=None):
        k, v = self.peek()
        if k == kind and (val is None or v == val):
            self.i += 1
            return True
        return False

    def expect(self, kind, val=None):
        k, v = self.peek()
        if not self.accept(kind, val):
            raise SyntaxException(f"expected {val or kind} but found '{v}'")
        return v

    # precedence climbing, lowest first
    def expression(self) -> Expression:
        return self.or_()

    def or_(self):
        left = self.and_()
        while self.accept("kw", "OR"):
            left = FunctionExpression(Function.OR, [left, self.and_()])
        return left

    def and_(self):
        left = self.compare()
        while self.accept("kw", "AND"):
            left = FunctionExpression(Function.AND, [left, self.compare()])
        return left

    def compare(self):
        left = self.add()
        while self.peek()[0] == "op" and self.peek()[1] in _COMPARISONS:
            op = self.take()[1]
            left = FunctionExpression(_COMPARISONS[op], [left, self.add()])
        return left

    def add(self):
        left = self.mul()
        while self.peek() in (("op", "+"), ("op", "-")):
            op = self.take()[1]
            left = FunctionExpression(Function.ADD if op == "+" else Function.SUB, [left, self.mul()])
        return left

    def mul(self):
        left = self.unary()
        while self.peek() in (("op", "*"), ("op", "/"), ("op", "%")):
            op = self.take()[1]
            f = {"*": Function.MUL, "/": Function.DIV, "%": Function.MOD}[op]
            left = FunctionExpression(f, [left, self.unary()])
        return left

    def unary(self):
        k, v = self.peek()
        if (k, v) in (("op", "-"), ("op", "+"), ("kw", "NOT")):
            self.take()
            operand = self.unary()
            f = {"-": Function.UNARY_MINUS, "+": Function.UNARY_PLUS, "NOT": Function.NOT}[v]
            if isinstance(operand, NumericLiteralExpression):      # ExpressionAstBuilder.kt:104-110
                if f == Function.UNARY_MINUS:
                    return NumericLiteralExpression(-operand.value)
                if f == Function.UNARY_PLUS:
                    return operand
            return FunctionExpression(f, [operand])
        return self.primary()

    def primary(self):
        k, v = self.take()
        if k in ("decimal", "integer"):
            return NumericLiteralExpression(float(v))               # :26-28 toDouble()
        if k == "kw" and v in ("TRUE", "FALSE"):
            return BooleanLiteralExpression(v == "TRUE")
        if k == "string":
            return StringLiteralExpression(v[1:-1].replace("''", "'"))
        if k == "kw" and v == "IF":
            c = self.expression(); self.expect("kw", "THEN")
            t = self.expression(); self.expect("kw", "ELSE")
            e = self.expression(); self.expect("kw", "END")
            return FunctionExpression(Function.IF, [c, t, e])
        if k == "op" and v == "(":
            e = self.expression()
            self.expect("op", ")")
            return e
        if k == "ident":
            if self.accept("op", "("):
                ops = [self.expression()]
                while self.accept("op", ","):
                    ops.append(self.expression())
                self.expect("op", ")")
                name = v.upper()
                if name in Function.__members__:
                    return FunctionExpression(Function[name], ops)
                if name in AggregationFunction.__members__:
                    return AggregationFunctionExpression(AggregationFunction[name], ops)
                raise SyntaxException(f"Unsupported function {name}")   # :62
            return IdentifierExpression(v)
        if k == "quoted":
            return IdentifierExpression(v[1:-1].replace('""', '"'))
        raise SyntaxException(f"unexpected token '{v}'")

    def identifier(self) -> str:
        k, v = self.take()
        if k == "ident":
            return v
        if k == "quoted":
            return v[1:-1].replace('""', '"')
        raise SyntaxException(f"expected identifier but found '{v}'")


def parseExpression(text: str) -> Expression:
    """parser/ParserHelper.kt:44-46"""
    p = _Parser(text)
    e = p.expression()
    p.expect("eof")
    return e


def parseQuery(text: str) -> Query:
    """parser/ParserHelper.kt:48-57"""
    p = _Parser(text)
    p.expect("kw", "SELECT")
    select = [p.expression()]
    while p.accept("op", ","):
        select.append(p.expression())
    p.expect("kw", "FROM")
    from_ = p.identifier()
    flt = p.expression() if p.accept("kw", "WHERE") else None
    order = None
    if p.accept("kw", "ORDER"):
        p.expect("kw", "BY")
        order = int(p.expect("integer"))
    p.expect("eof")
    return Query(tuple(select), from_, flt, order)
