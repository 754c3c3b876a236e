def box_area(boxes):
    """Area of xyxy boxes (public torchvision.ops.boxes.box_area definition)."""
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
