"""Eval metrics of the DocVQA / CORD tasks (behaviour of the reference's utils/metrics.py:3-25 and the JSONParseEvaluator of
utils/json_utils.py:115-317).  The reference leans on three packages that are not installable here -- `Levenshtein`, `nltk`
(edit_distance) and `zss` (Zhang-Shasha tree edit distance) -- so their published algorithms are restated below; parity with the
packages themselves is pinned by known-answer tests (tests/test_host_cpu.py: the zss README example, hand-computed ANLS / nTED cases)."""
from typing import Any, Callable, Dict, List, Union


def edit_distance(a, b) -> int:
    """Levenshtein distance (unit insert / delete / substitute), what Levenshtein.distance and nltk.edit_distance return"""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


# ---------------------------------------------------------------------------------------------- ANLS (ref utils/metrics.py)
def normalized_levenshtein(s1: str, s2: str) -> float:
    return edit_distance(s1, s2) / max(len(s1), len(s2))


def similarity_score(a_ij: str, o_q_i: str, tau: float = 0.5) -> float:
    nl = normalized_levenshtein(a_ij, o_q_i)
    return 1 - nl if nl < tau else 0


def average_normalized_levenshtein_similarity(ground_truth: List[List[str]], predicted_answers: List[str]) -> float:
    assert len(ground_truth) == len(predicted_answers), 'Length of ground_truth and predicted_answers must match.'
    total = 0.0
    for answers, pred in zip(ground_truth, predicted_answers):
        total += max(similarity_score(a, pred) for a in answers)
    return total / len(ground_truth)


# ---------------------------------------------------------------------------------------------- ordered tree edit distance
class Node:
    """labelled ordered tree node (the part of zss.Node the evaluator uses)"""

    def __init__(self, label: str, children=None):
        self.label = label
        self.children = list(children or [])

    def addkid(self, node: 'Node', before: bool = False) -> 'Node':
        if before:
            self.children.insert(0, node)
        else:
            self.children.append(node)
        return self

    @staticmethod
    def get_children(node: 'Node'):
        return node.children


def tree_edit_distance(a: Node, b: Node, insert_cost: Callable[[Node], float], remove_cost: Callable[[Node], float],
                       update_cost: Callable[[Node, Node], float]) -> float:
    """Zhang & Shasha (1989): minimum-cost sequence of node insertions, removals and relabelings that turns ordered tree a into b
    (zss.distance with the same three cost callbacks)."""
    def annotate(root):
        nodes, lmd = [], []                       # post-order nodes, index of each node's leftmost leaf descendant

        def walk(n):
            first = None
            for c in n.children:
                f = walk(c)
                if first is None:
                    first = f
            nodes.append(n)
            idx = len(nodes) - 1
            lmd.append(idx if first is None else first)
            return lmd[idx]
        walk(root)
        seen, keyroots = set(), []
        for i in range(len(nodes) - 1, -1, -1):   # keyroots: for every distinct leftmost leaf, the highest node that has it
            if lmd[i] not in seen:
                seen.add(lmd[i])
                keyroots.append(i)
        return nodes, lmd, sorted(keyroots)

    an, al, ak = annotate(a)
    bn, bl, bk = annotate(b)
    td = [[0.0] * len(bn) for _ in an]
    for i in ak:
        for j in bk:
            m, n = i - al[i] + 2, j - bl[j] + 2
            fd = [[0.0] * n for _ in range(m)]
            ioff, joff = al[i] - 1, bl[j] - 1
            for x in range(1, m):
                fd[x][0] = fd[x - 1][0] + remove_cost(an[x + ioff])
            for y in range(1, n):
                fd[0][y] = fd[0][y - 1] + insert_cost(bn[y + joff])
            for x in range(1, m):
                for y in range(1, n):
                    if al[i] == al[x + ioff] and bl[j] == bl[y + joff]:      # both prefixes are whole subtrees
                        fd[x][y] = min(fd[x - 1][y] + remove_cost(an[x + ioff]), fd[x][y - 1] + insert_cost(bn[y + joff]),
                                       fd[x - 1][y - 1] + update_cost(an[x + ioff], bn[y + joff]))
                        td[x + ioff][y + joff] = fd[x][y]
                    else:
                        p, q = al[x + ioff] - 1 - ioff, bl[y + joff] - 1 - joff
                        fd[x][y] = min(fd[x - 1][y] + remove_cost(an[x + ioff]), fd[x][y - 1] + insert_cost(bn[y + joff]),
                                       fd[p][q] + td[x + ioff][y + joff])
    return td[len(an) - 1][len(bn) - 1]


class JSONParseEvaluator:
    """n-TED (normalised tree edit distance) accuracy and field-level micro F1 of predicted vs ground-truth JSON
    (ref utils/json_utils.py:115-317, Donut's evaluator)"""

    @staticmethod
    def flatten(data: dict):
        out = []

        def _flatten(value, key=''):
            if type(value) is dict:
                for ck, cv in value.items():
                    _flatten(cv, f'{key}.{ck}' if key else ck)
            elif type(value) is list:
                for item in value:
                    _flatten(item, key)
            else:
                out.append((key, value))
        _flatten(data)
        return out

    @staticmethod
    def update_cost(node1: Node, node2: Node):
        l1, l2 = node1.label, node2.label
        leaf1, leaf2 = '<leaf>' in l1, '<leaf>' in l2
        if leaf1 and leaf2:
            return edit_distance(l1.replace('<leaf>', ''), l2.replace('<leaf>', ''))
        if not leaf1 and leaf2:
            return 1 + len(l2.replace('<leaf>', ''))
        if leaf1 and not leaf2:
            return 1 + len(l1.replace('<leaf>', ''))
        return int(l1 != l2)

    @staticmethod
    def insert_and_remove_cost(node: Node):
        label = node.label
        return len(label.replace('<leaf>', '')) if '<leaf>' in label else 1

    def normalize_dict(self, data: Union[Dict, List, Any]):
        if not data:
            return {}
        if isinstance(data, dict):
            new = {}
            for key in sorted(data.keys(), key=lambda k: (len(k), k)):
                value = self.normalize_dict(data[key])
                if value:
                    new[key] = value if isinstance(value, list) else [value]
            return new
        if isinstance(data, list):
            if all(isinstance(item, dict) for item in data):
                return [v for v in (self.normalize_dict(item) for item in data) if v]
            return [str(item).strip() for item in data if type(item) in {str, int, float} and str(item).strip()]
        return [str(data).strip()]

    def cal_f1(self, preds: List[dict], answers: List[dict]) -> float:
        tp, fn_or_fp = 0, 0
        for pred, answer in zip(preds, answers):
            pred, answer = self.flatten(self.normalize_dict(pred)), self.flatten(self.normalize_dict(answer))
            for fld in pred:
                if fld in answer:
                    tp += 1
                    answer.remove(fld)
                else:
                    fn_or_fp += 1
            fn_or_fp += len(answer)
        return tp / (tp + fn_or_fp / 2)

    def construct_tree_from_dict(self, data: Union[Dict, List], node_name: str = None) -> Node:
        node = Node('<root>' if node_name is None else node_name)
        if isinstance(data, dict):
            for key, value in data.items():
                node.addkid(self.construct_tree_from_dict(value, key))
        elif isinstance(data, list):
            if all(isinstance(item, dict) for item in data):
                for item in data:
                    node.addkid(self.construct_tree_from_dict(item, '<subtree>'))
            else:
                for item in data:
                    node.addkid(Node(f'<leaf>{item}'))
        else:
            raise Exception(data, node_name)
        return node

    def cal_acc(self, pred: dict, answer: dict) -> float:
        p = self.construct_tree_from_dict(self.normalize_dict(pred))
        a = self.construct_tree_from_dict(self.normalize_dict(answer))
        c = dict(insert_cost=self.insert_and_remove_cost, remove_cost=self.insert_and_remove_cost, update_cost=self.update_cost)
        empty = self.construct_tree_from_dict(self.normalize_dict({}))
        return max(0, 1 - tree_edit_distance(p, a, **c) / tree_edit_distance(empty, a, **c))
