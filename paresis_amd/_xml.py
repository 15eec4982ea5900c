"""Locating and reading the four XML configuration files (same schema as CodePython/xmlFiles/*.xml)."""
import os
from xml.dom import minidom

_PKG_XML = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xmlFiles")


def xml_dir(exp_dict=None):
    """Resolution order: exp_dict['xmlDir'], $PARESIS_XML_DIR, ./xmlFiles (the reference's cwd convention,
    Experiment.py:32), then the files shipped with this package."""
    if exp_dict is not None and exp_dict.get("xmlDir"):
        return exp_dict["xmlDir"]
    if os.environ.get("PARESIS_XML_DIR"):
        return os.environ["PARESIS_XML_DIR"]
    if os.path.isdir("xmlFiles"):
        return "xmlFiles"
    return _PKG_XML


def parse(directory, name):
    return minidom.parse(os.path.join(directory, name))


def text(node):
    return node.childNodes[0].nodeValue


def child_text(parent, tag):
    return text(parent.getElementsByTagName(tag)[0])


def has_child(parent, tag):
    return any(n.localName == tag for n in parent.childNodes)


def find_named(doc, element, name):
    for node in doc.documentElement.getElementsByTagName(element):
        if child_text(node, "name") == name:
            return node
    return None
