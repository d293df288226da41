"""ORACLE TOOLING -- runs ONLY in the build container (reads /root/reference).  Extracts the kinematic skeleton (link
names and, per joint, name / type / parent link / child link, all in file order) of the reference's robot URDFs into
tests/golden/urdf_skeletons.json.  Only names and the parent/child relation are kept -- no geometry, inertia, meshes or
any other text of the URDF files -- which is all the reference's graph construction reads (graphParser.py:97-148)."""
import json
import os
import xml.etree.ElementTree as ET

REF = "/root/reference/urdf_files"
ROBOTS = {
    "go1": "Go1/go1.urdf", "hyq": "HyQ/hyq.urdf", "a1": "A1/a1.urdf", "a1_quad_pruned": "A1-Quad/a1_pruned.urdf",
    "go2_quad": "Go2-Quad/go2.urdf", "mini_cheetah": "MiniCheetah/miniCheetah.urdf", "solo12_ori": "Solo_ori/solo12.urdf",
}


def skeleton(path):
    root = ET.parse(path).getroot()
    links = [l.attrib["name"] for l in root.findall("link")]
    joints = [[j.attrib["name"], j.attrib.get("type", ""), j.find("parent").attrib["link"], j.find("child").attrib["link"]]
              for j in root.findall("joint")]
    return {"links": links, "joints": joints}


if __name__ == "__main__":
    out = {}
    for name, rel in ROBOTS.items():
        p = os.path.join(REF, rel)
        if os.path.exists(p):
            out[name] = skeleton(p)
            print(name, len(out[name]["links"]), "links", len(out[name]["joints"]), "joints")
        else:
            print("missing", p)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "urdf_skeletons.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
